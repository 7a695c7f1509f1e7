"""GPU: the cross-lane 1024-point transform alone against numpy, both directions, with a per-position error map."""
import ctypes, sys, os
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch_cfd_amd as tc

def kmap():
    K = np.zeros((8, 128), dtype=int)
    for t in range(8):
        for j in range(128):
            w, l = j >> 6, j & 63
            lb = lambda b: (l >> b) & 1
            K[t, j] = ((w << 2) | (lb(1) << 1) | lb(0)) + 8 * ((lb(5) << 2) | (lb(4) << 1) | (t & 1)) \
                + 64 * ((lb(3) << 1) | lb(2)) + 256 * ((((t >> 2) & 1) << 1) | ((t >> 1) & 1))
    return K

dev = torch.device("cuda")
plan = tc.fft_plan(1024, torch.complex128, dev)
lib = tc._lib.load()
rng = np.random.default_rng(0)
S = 3
z = rng.standard_normal((S, 1024)) + 1j * rng.standard_normal((S, 1024))
K = kmap()
st = ctypes.c_void_p(torch.cuda.current_stream(dev).cuda_stream)
zin = torch.from_numpy(z).to(dev)
out = torch.empty_like(zin)
tc._lib.check(lib.tcfd_debug_xl_fft1024(plan.handle, zin.data_ptr(), out.data_ptr(), S, 1, st), "xl +1")
torch.cuda.synchronize()
o = out.cpu().numpy().reshape(S, 8, 128)
ref = np.fft.ifft(z, axis=-1) * 1024
got = np.zeros_like(ref)
for t in range(8):
    for j in range(128):
        got[:, K[t, j]] = o[:, t, j]
err = np.abs(got - ref) / np.abs(ref).max()
print("inverse max err", err.max())
if err.max() > 1e-12:
    bad = np.argwhere(err > 1e-12)
    print("bad count", len(bad), "of", err.size, "first", bad[:20].tolist())
    posbad = sorted({(t, j) for t in range(8) for j in range(128) if err[0, K[t, j]] > 1e-12})
    print("bad (t,j) positions (seq 0):", posbad[:64], len(posbad))
# forward
pin = np.zeros((S, 8, 128), dtype=complex)
pp = rng.standard_normal((S, 1024)) + 1j * rng.standard_normal((S, 1024))
for t in range(8):
    for j in range(128):
        pin[:, t, j] = pp[:, K[t, j]]
tin = torch.from_numpy(pin.reshape(S, 1024)).to(dev)
tc._lib.check(lib.tcfd_debug_xl_fft1024(plan.handle, tin.data_ptr(), out.data_ptr(), S, -1, st), "xl -1")
torch.cuda.synchronize()
f = out.cpu().numpy()
reff = np.fft.fft(pp, axis=-1)
errf = np.abs(f - reff) / np.abs(reff).max()
print("forward max err", errf.max())
if errf.max() > 1e-12:
    bad = np.argwhere(errf > 1e-12)
    print("bad count", len(bad), "first", bad[:20].tolist())
