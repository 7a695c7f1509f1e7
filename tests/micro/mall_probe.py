"""GPU: STREAM-style probe at working-set sizes inside and outside the 256 MB Infinity Cache."""
import ctypes, sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch_cfd_amd as tc
lib = tc._lib.load(); dev = torch.device("cuda")
st = ctypes.c_void_p(torch.cuda.current_stream(dev).cuda_stream)
for mb in (32, 64, 128, 192, 512, 1024):
    n = mb << 20
    a = torch.empty(n, dtype=torch.uint8, device=dev); b = torch.empty(n, dtype=torch.uint8, device=dev)
    res = []
    for mode, name, x in ((0, "copy", 2), (1, "read", 1), (2, "fill", 1)):
        t = ctypes.c_float(0)
        tc._lib.check(lib.tcfd_hbm_probe(a.data_ptr(), b.data_ptr(), n, mode, 30, ctypes.byref(t), st), "probe")
        res.append("%s %.2f TB/s" % (name, x * n / (t.value * 1e-3) / 1e12))
    print("%5d MB per buffer: " % mb + "  ".join(res))
