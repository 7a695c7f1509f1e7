cd $GRAFT_REPO_ROOT
python -m pytest tests/test_fno_gpu.py tests/test_training_kernels_gpu.py -m gpu -x -q 2>&1 | tail -4
CASES=200 python tests/micro/pw_tiles_fuzz.py 2>&1 | tail -2
ROWS=12 python tests/micro/train_step_ops.py 2>&1 | tail -14 | cut -c1-200
python - <<'PY' > gpurun_out/r05_sfno_c5_v3.json 2>gpurun_out/r05_sfno_c5_v3.err
import json, torch, bench
dev = torch.device('cuda:0')
print(json.dumps(bench.sfno_config5(dev, with_cpu=False), indent=1))
PY
python - <<'PY'
import json
d = json.load(open('gpurun_out/r05_sfno_c5_v3.json'))
print({k: v for k, v in d.items() if isinstance(v, (int, float))})
PY
