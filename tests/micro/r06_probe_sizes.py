"""STREAM-style probe (tcfd_hbm_probe: 16-byte lanes, grid-stride) at buffer sizes from cache-resident to HBM-sized:
what a plain fill / read / copy reaches when the buffer lives in the 256 MB Infinity Cache, against the 1 GiB figures bench.py
prints.  The solver's chunked step works on a 239 MB resident set; its passes are write-heavy (36 of 70 S per step)."""
import ctypes, json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import torch_cfd_amd as tc
lib = tc._lib.load()
dev = torch.device("cuda:0")
st = ctypes.c_void_p(torch.cuda.current_stream(dev).cuda_stream)
out = {}
for mb in (16, 32, 64, 96, 128, 192, 256, 512, 1024):
    n = mb << 20
    a = torch.empty(n, dtype=torch.uint8, device=dev); b = torch.empty(n, dtype=torch.uint8, device=dev)
    row = {}
    for mode, name, x in ((0, "copy", 2), (1, "read", 1), (2, "fill", 1)):
        t = ctypes.c_float(0)
        tc._lib.check(lib.tcfd_hbm_probe(a.data_ptr(), b.data_ptr(), n, mode, 20, ctypes.byref(t), st), "probe")
        row[name + "_GBps"] = round(x * n / (t.value * 1e-3) / 1e9, 1)
    out[f"{mb}MB"] = row
    print(mb, row, flush=True)
os.makedirs("gpurun_out", exist_ok=True)
json.dump(out, open("gpurun_out/r06_probe_sizes.json", "w"), indent=1)
