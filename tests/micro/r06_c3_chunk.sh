#!/bin/bash
run() { python tests/micro/r06_solver_ab.py --measure . 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('$LABEL', sorted(d['regions_ms_per_step'])[1], d['kernel_ms_per_step'])"; }
export AB_N=1024 AB_B=64 AB_STEPS=20 AB_DTYPE=f64
LABEL="default" run
for c in 2 3 4 5 6 8; do LABEL="chunk=$c" TCFD_CHUNK=$c run; done
for v in 1 2 3; do LABEL="rows_blocks_per_cu=$v" TCFD_ROWS_BLOCKS_PER_CU=$v run; done
LABEL="two_wg=0" TCFD_TWO_WG=0 run
LABEL="cols_xl=0" TCFD_COLS_XL=0 run
