# round 5, third GPU call: the FNO GPU suites on the pruned library (one tiled backward kernel for every width), timing with the
# launch no longer capped at 2048 partial rows, a first bench line
cd $GRAFT_REPO_ROOT
python -m pytest tests/test_fno_gpu.py tests/test_training_kernels_gpu.py -m gpu -q -x 2>&1 | tail -12 > gpurun_out/r05_run3_fno_tests.log
tail -12 gpurun_out/r05_run3_fno_tests.log
timeout 600 python tests/micro/pw_bwd_wide_timing.py 4 8 10 12 16 20 32 > gpurun_out/r05_pw_bwd_wide2.json 2> gpurun_out/r05_pw_bwd_wide2.err
python - <<'PY'
import json
for k, v in json.load(open("gpurun_out/r05_pw_bwd_wide2.json")).items(): print(k, v)
PY
python bench.py --no-c4 > gpurun_out/r05_run3_bench.json 2> gpurun_out/r05_run3_bench.err; tail -3 gpurun_out/r05_run3_bench.err
