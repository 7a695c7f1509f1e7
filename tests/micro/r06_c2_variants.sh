#!/bin/bash
for tree in ab/head .; do
for st in -1 0 1; do
  TCFD_SMALL_TILES=$st AB_DTYPE=f32 AB_N=256 AB_B=16 AB_STEPS=200 python tests/micro/r06_solver_ab.py --measure $tree 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('$tree small_tiles=$st', min(d['regions_ms_per_step']), d['kernel_ms_per_step'])"
done
for nt in 0 1; do
  TCFD_NT_PLANES=$nt AB_DTYPE=f32 AB_N=256 AB_B=16 AB_STEPS=200 python tests/micro/r06_solver_ab.py --measure $tree 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('$tree nt_planes=$nt', min(d['regions_ms_per_step']), d['kernel_ms_per_step'])"
done
done
