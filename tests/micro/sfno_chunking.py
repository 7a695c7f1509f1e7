"""GPU: does running the config-5 SFNO forward in cache-sized sample chunks pay?  (eager and CUDA-graph replayed)"""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from torch_cfd_amd import fno

dev = torch.device("cuda")
torch.manual_seed(0)
model = fno.SFNO(24, 24, 5, width=10, num_spectral_layers=4).to(dev).eval()
g = torch.Generator(device="cpu").manual_seed(0)
x = torch.randn(32, 256, 256, 10, generator=g).to(dev)

def timeit(fn, n=10):
    fn(); torch.cuda.synchronize()
    t = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t) / n * 1e3

with torch.no_grad():
    ref = model(x)
    print("whole batch       %.3f ms" % timeit(lambda: model(x)))
    for c in (16, 8, 4, 2, 1):
        def run():
            return torch.cat([model(x[i:i + c]) for i in range(0, 32, c)])
        out = run()
        assert torch.allclose(out, ref, rtol=1e-4, atol=1e-5)
        print("chunks of %2d eager %.3f ms" % (c, timeit(run)))
    for c in (8, 4, 2):
        xs = torch.empty(c, 256, 256, 10, device=dev)
        s = torch.cuda.Stream()
        with torch.cuda.stream(s):
            xs.copy_(x[:c]); model(xs); torch.cuda.synchronize()
            gr = torch.cuda.CUDAGraph()
            with torch.cuda.graph(gr, stream=s):
                ys = model(xs)
        outb = torch.empty_like(ref)
        def run_graph():
            for i in range(0, 32, c):
                xs.copy_(x[i:i + c]); gr.replay(); outb[i:i + c].copy_(ys)
        run_graph(); torch.cuda.synchronize()
        assert torch.allclose(outb, ref, rtol=1e-4, atol=1e-5)
        print("chunks of %2d graph %.3f ms" % (c, timeit(run_graph)))
