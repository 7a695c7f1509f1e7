import os, sys, torch
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/tests')
from test_ns2d_gpu import build_op, REAL, L
from conftest import rel_l2
from oracle import ns2d as O
dev = torch.device('cuda')
for n, tag in [(16, "f64"), (64, "f32"), (256, "f64"), (512, "f32"), (64, "f64"), (256, "f32")]:
    real = REAL[tag]
    w0 = torch.stack([torch.fft.rfft2(O.mcwilliams_vorticity(n, L, 4, s, real)) for s in range(2)]).to(dev)
    res = {}
    for flag in ("0", "1"):
        os.environ["TCFD_SPLIT"] = flag
        _, op = build_op(n, tag, "kolmogorov", dev)
        out, dwdt = op(w0, 1e-3, steps=3)
        res[flag] = (out, dwdt, op.explicit_terms(w0), op.residual(out, dwdt))
    print(n, tag, [rel_l2(a, b) for a, b in zip(res["0"], res["1"])])
