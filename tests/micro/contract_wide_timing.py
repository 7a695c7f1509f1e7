"""Contraction, adjoint and weight gradient at the config-5 grid for the widths the reference uses, kernel against kernel
(library events around each launch):  python tests/micro/contract_wide_timing.py 10 16 20 32
    contract: lanes (narrow) / per-mode product kernel (TCFD_CONTRACT_GEMM) / matrix-pipe kernel
    wgrad:    batch-split lanes kernel / per-mode product kernel (TCFD_WGRAD_GEMM)"""
import os, sys, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import bench
from torch_cfd_amd import fno

dev = torch.device("cuda:0")
widths = [int(x) for x in sys.argv[1:]] or [10, 16, 20, 32]
res = {}
for width in widths:
    b, modes = 32, (24, 24, 5)
    g = torch.Generator().manual_seed(0)
    vh = torch.view_as_complex(torch.randn(b, width, 48, 48, 5, 2, generator=g)).to(dev)
    w = [torch.randn(width, width, *modes, 2, generator=g).to(dev).requires_grad_(True) for _ in range(4)]
    bias = [torch.randn(*modes, 2, generator=g).to(dev).requires_grad_(True) for _ in range(4)]
    vr = vh.clone().requires_grad_(True)
    cot = torch.randn_like(vh)
    nbytes_c = (2 * vh.numel() + 4 * w[0].numel() // 2) * 8

    def step():
        out = fno._ContractFn.apply(vr, 0.5, modes, True, True, *w, *bias)
        torch.autograd.backward(out, cot)
    for tag, env in (("lanes_or_gemm", {}), ("gemm", {"TCFD_CONTRACT_LANES": 0, "TCFD_WGRAD_GEMM": 1}),
                     ("mfma_and_lanes_wgrad", {"TCFD_CONTRACT_LANES": 0, "TCFD_CONTRACT_GEMM": 0, "TCFD_WGRAD_GEMM": 0})):
        for k, v in env.items():
            os.environ[k] = str(v)
        t = bench.fno_kernel_times(step, dev, reps=10)
        res[f"w{width}_{tag}"] = {"contract_us": round(t["contract"]["avg_ms"] * 1e3, 1),
                                  "contract_GBps": round(nbytes_c / t["contract"]["avg_ms"] / 1e6),
                                  "wgrad_us": round(t["contract_wgrad"]["avg_ms"] * 1e3, 1),
                                  "wgrad_GBps": round(nbytes_c / t["contract_wgrad"]["avg_ms"] / 1e6)}
        for k in env:
            os.environ.pop(k)
print(json.dumps(res, indent=1))
