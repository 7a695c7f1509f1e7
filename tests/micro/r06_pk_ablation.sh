#!/bin/bash
# which group of packed fp32 forms costs the small-grid kernels (C2: 256^2 x 16) their 6 %?  one group switched off per tree
for tree in ab/head ab/noADD ab/noCMUL ab/noROT; do
  for cfg in "256 16 200" "128 8 200" "512 64 40"; do
    set -- $cfg
    AB_DTYPE=f32 AB_N=$1 AB_B=$2 AB_STEPS=$3 python tests/micro/r06_solver_ab.py --measure $tree 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('$tree', '$cfg', min(d['regions_ms_per_step']), d['kernel_ms_per_step'])"
  done
done
