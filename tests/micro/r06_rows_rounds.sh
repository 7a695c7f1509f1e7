run() { python tests/micro/r06_solver_ab.py --measure . 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('$LABEL', sorted(d['regions_ms_per_step'])[1], d['kernel_ms_per_step'])"; }
export AB_N=1024 AB_B=64 AB_STEPS=20 AB_DTYPE=f64
for per in 4 5 6 8 16; do LABEL="f64 1024 per_cu=$per" TCFD_ROWS_BLOCKS_PER_CU=$per run; done
export AB_N=512 AB_B=64 AB_STEPS=40 AB_DTYPE=f32
for per in 3 4 6 8 12; do LABEL="f32 512 per_cu=$per" TCFD_ROWS_BLOCKS_PER_CU=$per run; done
export AB_N=1024 AB_B=64 AB_STEPS=20 AB_DTYPE=f32
for per in 8 16; do LABEL="f32 1024 per_cu=$per" TCFD_ROWS_BLOCKS_PER_CU=$per run; done
