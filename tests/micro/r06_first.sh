#!/bin/bash
# round 6, first GPU call: the new bench record (read the way the driver reads it), the launch tests, the r04-vs-HEAD A/B
mkdir -p gpurun_out
python -m pytest tests/test_launch_gpu.py -x -q -m gpu 2>&1 | tail -5 > gpurun_out/r06_first_pytest.txt
python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/r06_first_bench.out 2> gpurun_out/r06_first_bench.err
echo "bench rc=$?" >> gpurun_out/r06_first_pytest.txt
python tests/micro/r06_solver_ab.py ab/r04 . > gpurun_out/r06_solver_ab.out 2> gpurun_out/r06_solver_ab.err
echo "ab rc=$?" >> gpurun_out/r06_first_pytest.txt
cat gpurun_out/r06_first_pytest.txt; tail -c 600 gpurun_out/r06_first_bench.out; tail -40 gpurun_out/r06_solver_ab.out
