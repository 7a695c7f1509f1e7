cd $GRAFT_REPO_ROOT
CASES=200 python tests/micro/pw_tiles_fuzz.py 2>&1 | tail -1
python -m pytest tests/test_fno_gpu.py -m gpu -x -q -k "pointwise or tiled or wide or widths_16 or golden or gelu or GELU" 2>&1 | tail -2
python tests/micro/pw_bwd_wide_timing.py 8 10 16 20 32 2>/dev/null | python -c "
import json,sys
d=json.load(sys.stdin)
print({k:v['ms'] for k,v in d.items() if k.endswith('tiles1')})"
