#!/bin/bash
# fp64 cross-lane row kernel: register / LDS trades that fit more waves per SIMD (TCFD_ROWS7_VARIANT), C3 headline region
run() { python tests/micro/r06_solver_ab.py --measure . 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('$LABEL', sorted(d['regions_ms_per_step'])[1], d['kernel_ms_per_step'])"; }
export AB_N=1024 AB_B=64 AB_STEPS=20 AB_DTYPE=f64
for v in 0 1 2 3 4 0 1 2 3 4; do LABEL="variant=$v" TCFD_ROWS7_VARIANT=$v run; done
for v in 1 2 3 4; do for per in 4 5 6; do LABEL="variant=$v per_cu=$per" TCFD_ROWS7_VARIANT=$v TCFD_ROWS_BLOCKS_PER_CU=$per run; done; done
