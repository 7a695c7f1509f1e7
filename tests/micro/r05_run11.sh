# lanes contraction + batch-split weight gradient: whole FNO GPU suite, fuzz, wide-width contraction timing, SFNO lines of the bench
cd $GRAFT_REPO_ROOT
python -m pytest tests/test_fno_gpu.py tests/test_training_kernels_gpu.py -m gpu -x -q 2>&1 | tail -3
python tests/micro/contract_timing.py 16 20 32 2>/dev/null | python -c "
import json,sys
d=json.load(sys.stdin)
for k,v in d.items():
    print(k, v if 'us' not in v else (v['us'], v['GBps']))
"
python - <<'PY' 2>&1 | tail -30
import json, torch, bench
dev = torch.device('cuda:0')
r = bench.sfno_config5(dev, with_cpu=False)
print(json.dumps(r, indent=1)[:6000])
PY
