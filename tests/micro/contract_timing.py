"""The 4-corner contraction, its adjoint and its weight gradient at the config-5 shape (b 32, width 10, modes 24 x 24 x 5), timed by
the library's own events around each launch (tcfd_fno_profile_begin / _end):  python tests/micro/contract_timing.py
    lanes kernel (TCFD_CONTRACT_LANES=1, default for narrow fp32 layers) with TCFD_CONTRACT_BG slices of the batch, against the matrix-pipe kernel (TCFD_CONTRACT_LANES=0) and the plain kernel (use_mfma=False)."""
import os, sys, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import bench
from torch_cfd_amd import fno

dev = torch.device("cuda:0")
widths = [int(x) for x in sys.argv[1:]] or [10]
res = {}
for width in widths:
    b, ci, co, modes = 32, width, width, (24, 24, 5)
    mx, my, mt = modes
    g = torch.Generator().manual_seed(0)
    vh = torch.view_as_complex(torch.randn(b, ci, 2 * mx, 2 * my, mt, 2, generator=g)).to(dev)
    w = [torch.view_as_complex(torch.randn(ci, co, *modes, 2, generator=g)).to(dev) for _ in range(4)]
    bias = [torch.view_as_complex(torch.randn(*modes, 2, generator=g)).to(dev) for _ in range(4)]
    ref = fno.hip_contract(vh, w, bias, 0.5, modes, use_mfma=False)
    nbytes = (vh.numel() * 2 + sum(x.numel() for x in w)) * 8

    def timed(tag, **env):
        for k, v in env.items():
            os.environ[k] = str(v)
        out = fno.hip_contract(vh, w, bias, 0.5, modes)
        err = (torch.linalg.norm(torch.view_as_real(out - ref)) / torch.linalg.norm(torch.view_as_real(ref))).item()
        t = bench.fno_kernel_times(lambda: fno.hip_contract(vh, w, bias, 0.5, modes), dev, reps=20)
        k = t.get("contract", next(iter(t.values())))
        res[f"w{width}_{tag}"] = {"us": round(k["avg_ms"] * 1e3, 1), "GBps": round(nbytes / k["avg_ms"] / 1e6, 0), "rel_err_vs_plain": err}
        for k_ in env:
            os.environ.pop(k_)

    timed("mfma", TCFD_CONTRACT_LANES=0)
    for bg in (0, 2, 4, 8, 16):
        timed(f"lanes_bg{bg}", TCFD_CONTRACT_LANES=1, TCFD_CONTRACT_BG=bg)
    # the training side: adjoint + weight gradient through autograd of the contraction alone
    vr = vh.clone().requires_grad_(True)
    wr = [torch.view_as_real(x).clone().requires_grad_(True) for x in w]
    br = [torch.view_as_real(x).clone().requires_grad_(True) for x in bias]
    cot = torch.randn_like(ref)

    def step():
        out = fno._ContractFn.apply(vr, 0.5, modes, True, True, *wr, *br)
        torch.autograd.backward(out, cot)
    for lanes in (0, 1):
        os.environ["TCFD_CONTRACT_LANES"] = str(lanes)
        t = bench.fno_kernel_times(step, dev, reps=10)
        res[f"w{width}_train_lanes{lanes}"] = {k: {"n": v["launches"], "us": round(v["avg_ms"] * 1e3, 1)} for k, v in t.items()}
    os.environ.pop("TCFD_CONTRACT_LANES")
print(json.dumps(res, indent=1))
