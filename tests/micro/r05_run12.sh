cd $GRAFT_REPO_ROOT
python -m pytest tests/test_fno_gpu.py -m gpu -x -q -k "contraction" 2>&1 | tail -2
python - <<'PY' 2>/dev/null
import os, json, torch, bench
from torch_cfd_amd import fno
dev = torch.device('cuda:0')
for width in (14, 16, 20, 24, 32):
    b, modes = 32, (24, 24, 5)
    g = torch.Generator().manual_seed(0)
    vh = torch.view_as_complex(torch.randn(b, width, 48, 48, 5, 2, generator=g)).to(dev)
    w = [torch.view_as_complex(torch.randn(width, width, *modes, 2, generator=g)).to(dev) for _ in range(4)]
    ref = fno.hip_contract(vh, w, None, 1.0, modes, use_mfma=False)
    for nm in (8, 4, 0):
        os.environ["TCFD_CONTRACT_NM"] = str(nm)
        out = fno.hip_contract(vh, w, None, 1.0, modes)
        err = float((out - ref).abs().max() / ref.abs().max())
        t = bench.fno_kernel_times(lambda: fno.hip_contract(vh, w, None, 1.0, modes), dev, reps=20)
        print(width, "NM", nm, round(t["contract"]["avg_ms"] * 1e3, 1), "us", "err", err)
PY
python - <<'PY' > gpurun_out/r05_sfno_c5_v2.json 2>gpurun_out/r05_sfno_c5_v2.err
import json, torch, bench
dev = torch.device('cuda:0')
print(json.dumps(bench.sfno_config5(dev, with_cpu=False), indent=1))
PY
python - <<'PY'
import json
d = json.load(open('gpurun_out/r05_sfno_c5_v2.json'))
def walk(x, pre=''):
    for k, v in x.items():
        if isinstance(v, dict):
            if pre.count('.') < 1: walk(v, pre + k + '.')
        elif isinstance(v, (int, float)): print(pre + k, v)
walk(d)
PY
