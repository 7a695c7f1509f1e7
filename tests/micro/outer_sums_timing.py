import os, sys, ctypes, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from torch_cfd_amd import _lib
lib = _lib.load(); dev = torch.device("cuda:0")
b, C, co, P = 32, 10, 10, 655360
dy = torch.randn(b, co, P, device=dev); x1 = torch.randn(b, P, device=dev); pe = torch.randn(C, P, device=dev)
st = ctypes.c_void_p(torch.cuda.current_stream(dev).cuda_stream)
for wps in (64, 128, 256, 512):
    tiles = torch.empty(wps, b, 256, device=dev)
    f = lambda: lib.tcfd_fno_sample_outer_sums(dy.data_ptr(), x1.data_ptr(), pe.data_ptr(), tiles.data_ptr(), b, C, co, P, wps, st)
    f(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10): f()
    e1.record(); torch.cuda.synchronize()
    print(wps, round(e0.elapsed_time(e1) / 10 * 1e3, 1), "us")
