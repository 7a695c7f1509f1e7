"""Where does a torch-autograd training step of the SFNO pointwise blocks spend its time on MI355X?
Times fwd+bwd of (a) nn.Conv3d 1x1x1, (b) the same map as a channel matmul, (c) GroupNorm(1 group)."""
import torch, torch.nn as nn, torch.nn.functional as F, json
dev = torch.device("cuda:0")
b, C, X, Y, T = 32, 10, 256, 256, 10

def timeit(fn, n=3):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return round(e0.elapsed_time(e1) / n, 2)

x = torch.randn(b, C, X, Y, T, device=dev, requires_grad=True)
res = {}
conv = nn.Conv3d(C, 4 * C, 1).to(dev)
def f_conv():
    y = conv(x); y.sum().backward()
res["conv3d_10to40_fwd_bwd_ms"] = timeit(f_conv, 1)
W = torch.randn(4 * C, C, device=dev, requires_grad=True); bb = torch.randn(4 * C, device=dev, requires_grad=True)
def f_mm():
    y = torch.matmul(W, x.reshape(b, C, -1)) + bb[None, :, None]; y.sum().backward()
res["matmul_10to40_fwd_bwd_ms"] = timeit(f_mm)
def f_ein():
    y = torch.einsum("oc,bcp->bop", W, x.reshape(b, C, -1)); y.sum().backward()
res["einsum_10to40_fwd_bwd_ms"] = timeit(f_ein)
gn = nn.GroupNorm(1, C).to(dev)
def f_gn():
    y = gn(x); y.sum().backward()
res["groupnorm_fwd_bwd_ms"] = timeit(f_gn, 1)
def f_relu_block():
    xr = x.reshape(b, C, -1)
    h = torch.relu(torch.matmul(W, xr) + bb[None, :, None])
    o = torch.matmul(W.t(), h)
    o.sum().backward()
res["ffn_matmul_fwd_bwd_ms"] = timeit(f_relu_block)
print(json.dumps(res))
