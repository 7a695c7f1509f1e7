#!/bin/bash
# fp32 row pass: workgroups per CU of the persistent grid, packed (.) and scalar (ab/head) builds
mkdir -p gpurun_out
for cfg in "512 64 f32 40" "1024 64 f32 20"; do
  set -- $cfg
  for tree in ab/head .; do
    for per in 1 2 3 4; do
      TCFD_ROWS_BLOCKS_PER_CU=$per AB_N=$1 AB_B=$2 AB_DTYPE=$3 AB_STEPS=$4 python tests/micro/r06_solver_ab.py --measure $tree 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('$cfg', '$tree', 'per_cu=$per', min(d['regions_ms_per_step']), d['kernel_ms_per_step'])"
    done
  done
done 2>&1 | tee gpurun_out/r06_rows_sweep.txt
