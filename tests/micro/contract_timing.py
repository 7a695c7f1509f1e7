"""Timing of the 4-corner contraction at the config-5 shape, 8 against 16 modes per workgroup: python tests/micro/contract_timing.py"""
import os, sys, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from torch_cfd_amd import fno

dev = torch.device("cuda:0")
b, ci, co, modes = 32, 10, 10, (24, 24, 5)
mx, my, mt = modes
g = torch.Generator().manual_seed(0)
vh = torch.view_as_complex(torch.randn(b, ci, 2 * mx, 2 * my, mt, 2, generator=g)).to(dev)
w = [torch.view_as_complex(torch.randn(ci, co, *modes, 2, generator=g)).to(dev) for _ in range(4)]
res = {}
ref = fno.hip_contract(vh, w, None, 1.0, modes, use_mfma=False)
for nm in (8, 16):
    os.environ["TCFD_CONTRACT_NM"] = str(nm)
    out = fno.hip_contract(vh, w, None, 1.0, modes)
    err = (torch.linalg.norm(torch.view_as_real(out - ref)) / torch.linalg.norm(torch.view_as_real(ref))).item()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    best = 1e9
    for _ in range(5):
        e0.record()
        for _ in range(20):
            fno.hip_contract(vh, w, None, 1.0, modes)
        e1.record(); torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) / 20)
    res[f"nm{nm}"] = {"ms_incl_host": round(best, 4), "rel_err_vs_valu": err}
print(json.dumps(res))
