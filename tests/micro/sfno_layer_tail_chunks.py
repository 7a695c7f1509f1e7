"""GPU experiment: the tail of one SFNO hidden layer -- inverse t/y transform -> fused pointwise block -> forward t/y
transform of the NEXT layer -- on the whole batch vs in sample chunks whose x1 / out tensors stay in the Infinity
Cache.  Both under CUDA-graph replay (no host overhead)."""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from torch_cfd_amd import fno

dev = torch.device("cuda"); torch.manual_seed(0)
B, C, X, Y, T, modes = 32, 10, 256, 256, 10, (24, 24, 5)
mlp = fno.PointwiseFFN(C, C, 4 * C, "ReLU").to(dev); w = torch.nn.Conv3d(C, C, 1).to(dev); act = torch.nn.ReLU()
v = torch.randn(B, C, X, Y, T, device=dev)
with torch.no_grad():
    vh0, plan = fno.hip_truncated_rfftn(v, modes)
    O = (vh0 * 0.5).contiguous()
    out = torch.empty_like(v); vh_next = torch.empty_like(vh0)

    def tail(sl):
        x1 = fno.hip_truncated_irfftn(O[sl], plan, T)
        o = fno.hip_pointwise(x1, mlp.linear1, mlp.activation, mlp.linear2, skip=v[sl], skip_conv=w, act2=act)
        out[sl].copy_(o)
        vh, _ = fno.hip_truncated_rfftn(o, modes)
        vh_next[sl].copy_(vh)

    def run(c):
        for i in range(0, B, c):
            tail(slice(i, i + c))

    def graphed(c):
        s = torch.cuda.Stream()
        with torch.cuda.stream(s):
            run(c); torch.cuda.synchronize()
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g, stream=s):
                run(c)
        return g

    ref = None
    for c in (32, 16, 8, 4, 2, 1):
        g = graphed(c)
        g.replay(); torch.cuda.synchronize()
        if ref is None:
            ref = (out.clone(), vh_next.clone())
        else:
            assert torch.equal(out, ref[0]) and torch.equal(vh_next, ref[1])
        t0 = time.perf_counter()
        for _ in range(10): g.replay()
        torch.cuda.synchronize()
        print("chunk %2d: %.3f ms per layer tail" % (c, (time.perf_counter() - t0) / 10 * 1e3))
