import ctypes, sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from torch_cfd_amd import fno, _lib
dev = torch.device("cuda:0")
n, b, nt = int(sys.argv[1]), 1, 2
g = torch.Generator().manual_seed(n + nt)
x = torch.randn(b, n, n, nt, generator=g).to(dev); y = (x + 0.3 * torch.randn(b, n, n, nt, generator=g).to(dev))
loss = fno.SobolevLoss(n_grid=n, norm_order=0, relative=True).to(dev)
w2 = loss._half_spectrum_weights(dev, torch.float32)
lib = _lib.load(); plan = ctypes.c_void_p()
_lib.check(lib.tcfd_loss_plan_create(ctypes.byref(plan), n, 0))
need = lib.tcfd_loss_workspace_bytes(plan, b, nt, 2)
ws = torch.zeros(need, dtype=torch.uint8, device=dev); out = torch.empty((), device=dev); sums = torch.zeros(2, b, nt, dtype=torch.float64, device=dev)
_lib.check(lib.tcfd_sobolev_loss(plan, x.data_ptr(), y.data_ptr(), w2.data_ptr(), b, nt, 2, 1, 1, 1, 1, out.data_ptr(), sums.data_ptr(), ws.data_ptr(), need, None))
torch.cuda.synchronize()
def ref(z):
    zh = torch.fft.rfft2(z.double().permute(0, 3, 1, 2))
    return ((zh.real**2 + zh.imag**2) * w2.double()).sum(dim=(-2, -1)), zh
r0, dh = ref(x - y); r1, yh = ref(y)
print("sums", sums.cpu().tolist(), "ref", r0.cpu().tolist(), r1.cpu().tolist())
# the half-spectrum planes of pass 1: (f, b, t, i, ldk) complex64 ; compare with the row transform
ldk = ((n // 2 + 1) + 15) // 16 * 16
planes = ws[: 2 * b * nt * n * ldk * 8].view(torch.complex64).reshape(2, b, nt, n, ldk)[..., : n // 2 + 1]
rowD = torch.fft.rfft((x - y).double().permute(0, 3, 1, 2), dim=-1); rowY = torch.fft.rfft(y.double().permute(0, 3, 1, 2), dim=-1)
for f, rr in ((0, rowD), (1, rowY)):
    e = (planes[f].to(torch.complex128) - rr).abs()
    print("field", f, "row-pass max err", float(e.max()), "rel", float(e.max() / rr.abs().max()), "worst col", int(e.amax(dim=(0, 1, 2)).argmax()), "worst row", int(e.amax(dim=(0, 1, 3)).argmax()))
