#!/bin/bash
# packed fp32 + fp32 variants as defaults: quick A/B per config, then the full GPU suite, then the bench
mkdir -p gpurun_out
for tree in ab/head .; do
for cfg in "256 16 f32 200" "128 1 f64 200" "512 64 f32 40" "1024 64 f32 20" "1024 64 f64 20"; do
  set -- $cfg
  AB_DTYPE=$3 AB_N=$1 AB_B=$2 AB_STEPS=$4 python tests/micro/r06_solver_ab.py --measure $tree 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('$tree', '$cfg', min(d['regions_ms_per_step']), round(1e3/min(d['regions_ms_per_step']),1), d['kernel_ms_per_step'])"
done; done 2>&1 | tee gpurun_out/r06_third_ab.txt
python -m pytest tests -x -q -m gpu 2>&1 | tail -8 > gpurun_out/r06_third_pytest.txt
cat gpurun_out/r06_third_pytest.txt
python bench.py > gpurun_out/r06_third_bench.out 2> gpurun_out/r06_third_bench.err; echo "bench rc=$?"
tail -c 900 gpurun_out/r06_third_bench.out
