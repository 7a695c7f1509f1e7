#!/bin/bash
# round 6, second GPU call: the new parity tests; kernel stats + counters of the fp32 solver instantiations
mkdir -p gpurun_out
python -m pytest tests/test_ns2d_gpu.py tests/test_fno_gpu.py -x -q -m gpu -k "legacy_crank or backdiff or kolmogorov_dataset or second_order or config4_dataset or sobolev" 2>&1 | tail -8 > gpurun_out/r06_second_pytest.txt
bash tests/prof.sh r06_c4_f32 --n 512 --batch 64 --dtype f32 --regions 1 --preheat 0 > /dev/null 2>&1
bash tests/prof.sh r06_c4_f64 --n 512 --batch 64 --dtype f64 --regions 1 --preheat 0 > /dev/null 2>&1
bash tests/prof.sh r06_c2 --n 256 --batch 16 --dtype f32 --fused-steps --steps 20 --regions 1 --preheat 0 > /dev/null 2>&1
bash tests/prof.sh r06_c3_f32 --n 1024 --batch 64 --dtype f32 --regions 1 --preheat 0 > /dev/null 2>&1
cat gpurun_out/r06_second_pytest.txt
for t in r06_c4_f32 r06_c4_f64 r06_c2 r06_c3_f32; do echo "=== $t"; head -12 gpurun_out/prof_$t/summary.txt; done
