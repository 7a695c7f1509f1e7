# round 5, sixth GPU call: two groups of loads in flight per wave in the tile kernels
cd $GRAFT_REPO_ROOT
python -m pytest tests/test_fno_gpu.py -m gpu -x -q -k "wide_forward or pointwise_backward_kernel or tiled_backward" 2>&1 | tail -3
timeout 600 python tests/micro/pw_bwd_wide_timing.py 4 8 10 12 16 20 > gpurun_out/r05_pw_bwd_wide3.json 2>/dev/null
python - <<'PY'
import json
for k, v in json.load(open("gpurun_out/r05_pw_bwd_wide3.json")).items():
    if k.endswith("tiles1"): print(k, v)
PY
TCFD_PW_FWD_TILES=2 python - <<'PY'
import os, sys, json, torch, torch.nn as nn
sys.path.insert(0, os.getcwd())
from torch_cfd_amd import fno
dev = torch.device("cuda:0")
res = {}
for W in (16, 20, 24, 32):
    mlp = fno.PointwiseFFN(W, W, 4 * W, "ReLU").to(dev); w = nn.Conv3d(W, W, 1).to(dev); a = nn.ReLU()
    x1 = torch.randn(32, W, 256, 256, 10, device=dev); v = torch.randn_like(x1)
    for flag in ("2", "0"):
        os.environ["TCFD_PW_FWD_TILES"] = flag
        with torch.no_grad():
            f = lambda: fno.hip_pointwise(x1, mlp.linear1, a, mlp.linear2, skip=v, skip_conv=w, act2=a)
            f(); torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(5): f()
            e1.record(); torch.cuda.synchronize()
            res[f"w{W}_tiles{flag}"] = round(e0.elapsed_time(e1) / 5, 3)
    del x1, v
print(json.dumps(res))
PY
