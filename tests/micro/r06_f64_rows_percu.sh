#!/bin/bash
for cfg in "512 64 40" "256 64 100" "128 64 200" "768 64 20"; do
  set -- $cfg
  for per in 0 1 2 3; do
    TCFD_ROWS_BLOCKS_PER_CU=$per AB_DTYPE=f64 AB_N=$1 AB_B=$2 AB_STEPS=$3 python tests/micro/r06_solver_ab.py --measure . 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('f64 $cfg per_cu=$per', min(d['regions_ms_per_step']), d['kernel_ms_per_step'])"
  done
done
