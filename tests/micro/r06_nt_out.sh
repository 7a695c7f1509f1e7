#!/bin/bash
run() { python tests/micro/r06_solver_ab.py --measure . 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('$LABEL', sorted(d['regions_ms_per_step'])[1], d['kernel_ms_per_step'])"; }
for cfg in "1024 64 f64 20" "512 64 f64 40" "512 64 f32 40" "1024 64 f32 20" "256 16 f32 200"; do
  set -- $cfg
  export AB_N=$1 AB_B=$2 AB_DTYPE=$3 AB_STEPS=$4
  for v in 0 1 0 1; do LABEL="$cfg nt_out=$v" TCFD_NT_OUT=$v run; done
done
