# end-of-round evidence (round 6): GPU tests, default bench (read the way the driver reads it), rocprofv3 kernel stats + PMC of the headline
# and of the fp32 solver configs, traffic.json, SFNO PMC table, SFNO training-step kernel stats, torch ops left in a training step
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
python -m pytest tests -m gpu -q 2>&1 | tail -6 > gpurun_out/r06_final_tests.log
python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/r06_final_bench.json 2> gpurun_out/r06_final_bench.err
cp gpurun_out/bench_detail.json gpurun_out/r06_final_bench_detail.json
python - <<'PY' > gpurun_out/r06_final_bench_parse.txt
import json, sys
sys.path.insert(0, "tests")
from test_bench_record import read_like_the_driver
rec = read_like_the_driver(open("gpurun_out/r06_final_bench.json").read(), open("gpurun_out/r06_final_bench.err").read())
print("parsed like the driver:", len(open("gpurun_out/r06_final_bench.json").read()), "bytes;", rec["metric"], rec["value"], rec["ms_per_step"], rec["roofline"]["frac"], rec["cpu_baseline"]["value"])
PY
bash tests/prof.sh r06_final --regions 1 --preheat 2 > gpurun_out/r06_final_prof.log 2>&1
python tests/prof_traffic.py gpurun_out/prof_r06_final 1024 64 f64 16 "r06_final (round-6 build; fp64 solver kernels are round 4's)" > gpurun_out/r06_final_traffic.log 2>&1
cp profiles/traffic.json gpurun_out/r06_traffic.json
bash tests/prof.sh r06_c4_f32 --n 512 --batch 64 --dtype f32 --regions 1 --preheat 2 > /dev/null 2>&1
bash tests/prof.sh r06_c4_f64 --n 512 --batch 64 --dtype f64 --regions 1 --preheat 2 > /dev/null 2>&1
bash tests/prof.sh r06_c2 --n 256 --batch 16 --dtype f32 --fused-steps --steps 20 --regions 1 --preheat 2 > /dev/null 2>&1
bash tests/prof_sfno.sh r06 > gpurun_out/r06_final_sfno_pmc.log 2>&1
cd $GRAFT_REPO_ROOT
find gpurun_out/prof_sfno_r06 -type f \( -name '*kernel_trace.csv' -o -name '*.db' -o -name '*agent_info.csv' \) -delete
find gpurun_out/prof_sfno_r06 -type f -name '*counter_collection.csv' -size +6M -delete
TRAIN=1 ONLY_TRAIN=1 ROWS=40 bash tests/micro/sfno_profile.sh > gpurun_out/r06_final_sfno_train.txt 2>&1
cd $GRAFT_REPO_ROOT
rm -rf gpurun_out/prof_sfno2
ROWS=40 python tests/micro/train_step_ops.py 2>&1 | tail -42 > gpurun_out/r06_final_train_step_ops.txt
cat gpurun_out/r06_final_tests.log; cat gpurun_out/r06_final_bench_parse.txt; cat gpurun_out/r06_final_traffic.log; du -sh gpurun_out
