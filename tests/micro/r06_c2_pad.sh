#!/bin/bash
# C2 (256^2 x 16 fp32): does capping the column workgroups per CU (extra dynamic LDS) bring back the scalar build's times?
for pad in 0 8192 16384 24576 40960 65536; do
  for cfg in "256 16 200"; do
    set -- $cfg
    TCFD_COLS_LDS_PAD=$pad AB_DTYPE=f32 AB_N=$1 AB_B=$2 AB_STEPS=$3 python tests/micro/r06_solver_ab.py --measure . 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('pad=$pad', '$cfg', min(d['regions_ms_per_step']), d['kernel_ms_per_step'])"
  done
done
for cfg in "512 64 40" "1024 64 20"; do
  set -- $cfg
  AB_DTYPE=f32 AB_N=$1 AB_B=$2 AB_STEPS=$3 python tests/micro/r06_solver_ab.py --measure . 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('defaults', '$cfg', min(d['regions_ms_per_step']), d['kernel_ms_per_step'])"
done
