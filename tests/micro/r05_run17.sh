cd $GRAFT_REPO_ROOT
python -m pytest tests/test_fno_gpu.py -m gpu -x -q -k "contraction" 2>&1 | tail -3
echo KC4; python tests/micro/contract_wide_timing.py 14 16 20 24 32 2>/dev/null | python -c "
import json,sys
for k,v in json.load(sys.stdin).items(): print(k,v)"
echo KC2; TCFD_GEMM_KC=2 python tests/micro/contract_wide_timing.py 14 16 20 24 32 2>/dev/null | python -c "
import json,sys
for k,v in json.load(sys.stdin).items():
    if 'mfma' not in k: print(k,v)"
