#!/bin/bash
# packed fp32 complex arithmetic (tcfd_fft.hpp, TCFD_PK32): parity tests, then A/B against the HEAD build (ab/head) per config
mkdir -p gpurun_out
python -m pytest tests/test_ns2d_gpu.py tests/test_fno_gpu.py tests/test_training_kernels_gpu.py -x -q -m gpu 2>&1 | tail -8 > gpurun_out/r06_pk32_pytest.txt
cat gpurun_out/r06_pk32_pytest.txt
for cfg in "512 64 f32 40" "1024 64 f32 20" "256 16 f32 200" "512 64 f64 40"; do
  set -- $cfg
  AB_N=$1 AB_B=$2 AB_DTYPE=$3 AB_STEPS=$4 AB_REPS=2 AB_OUT=gpurun_out/r06_pk32_ab_n$1_$3.json python tests/micro/r06_solver_ab.py ab/head . 2>/dev/null | python -c "
import json,sys
d=json.load(sys.stdin)
for t,v in d.items(): print('$cfg', t, v['median_ms_per_step'], v['steps_per_s_median'], v['kernel_ms_per_step_median'])"
done
