cd $GRAFT_REPO_ROOT
python -m pytest tests/test_fno_gpu.py -m gpu -x -q -k "contraction" 2>&1 | tail -8
python tests/micro/contract_wide_timing.py 10 14 16 20 24 32 > gpurun_out/r05_contract_wide_timing.json 2>gpurun_out/r05_contract_wide_timing.err
python - <<'PY'
import json
for k, v in json.load(open('gpurun_out/r05_contract_wide_timing.json')).items(): print(k, v)
PY
tail -3 gpurun_out/r05_contract_wide_timing.err
