cd $GRAFT_REPO_ROOT
python -m pytest tests/test_launch_gpu.py -x -q -m gpu -k "two_ranks or c4" 2>&1 | tail -15 > gpurun_out/r04_run2_tests.log
python -m pytest tests/test_ns2d_gpu.py -x -q -m gpu -k "second_order or fused_vjp or require_grad" 2>&1 | tail -5 >> gpurun_out/r04_run2_tests.log
cat gpurun_out/r04_run2_tests.log
python bench.py > gpurun_out/r04_bench_a.json 2> gpurun_out/r04_bench_a.err
tail -3 gpurun_out/r04_bench_a.err
python - <<'PY'
import json
d = json.load(open("gpurun_out/r04_bench_a.json"))
print("value", d["value"], "ms", d["ms_per_step"])
print("proxy", json.dumps(d["strong_scaling_proxy"]))
print("c4", json.dumps(d["c4_ensemble"]))
print("other", json.dumps(d["other_configs"]))
s = d["sfno_config5"]; print("sfno", {k: s[k] for k in s if k != "roofline"}); print(json.dumps(s["roofline"]))
print("cpu", json.dumps(d["cpu_baseline"]))
PY
