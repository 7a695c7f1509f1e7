#!/bin/bash
# fp32 solver variants of round 6 (packed build): 8-column cross-lane column tiles, cross-lane row kernel at 1024
mkdir -p gpurun_out
run() {  # label, env..., then n B dtype steps
  python tests/micro/r06_solver_ab.py --measure . 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('$LABEL', min(d['regions_ms_per_step']), d['kernel_ms_per_step'])"
}
export AB_DTYPE=f32
for c8 in 0 4 6; do
  LABEL="512x64 cols8=$c8" TCFD_F32_COLS8=$c8 AB_N=512 AB_B=64 AB_STEPS=40 run
done
for c8 in 0 4 6; do
  for rv in "0 4" "7 4" "7 2"; do
    set -- $rv
    LABEL="1024x64 cols8=$c8 rows_v=$1 minw=$2" TCFD_F32_COLS8=$c8 TCFD_ROWS_V=$1 TCFD_ROWS7_MINW=$2 AB_N=1024 AB_B=64 AB_STEPS=20 run
  done
done
for per in 0 1 2 4 8; do
  LABEL="256x16 per_cu=$per" TCFD_ROWS_BLOCKS_PER_CU=$per AB_N=256 AB_B=16 AB_STEPS=200 run
  LABEL="256x64 per_cu=$per" TCFD_ROWS_BLOCKS_PER_CU=$per AB_N=256 AB_B=64 AB_STEPS=100 run
done
