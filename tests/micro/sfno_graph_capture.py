import os, sys, json, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from torch_cfd_amd import fno
dev = torch.device("cuda:0")
torch.manual_seed(0)
model = fno.SFNO(24, 24, 5, width=10, num_spectral_layers=4).to(dev).eval()
x = torch.randn(32, 256, 256, 10, device=dev)
def timeit(fn, n=10):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n
with torch.no_grad():
    y_ref = model(x)
    t_plain = timeit(lambda: model(x))
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        for _ in range(3): model(x)
    torch.cuda.current_stream().wait_stream(s)
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        y = model(x)
    g.replay(); torch.cuda.synchronize()
    err = ((y - y_ref).norm() / y_ref.norm()).item()
    t_graph = timeit(lambda: g.replay())
print(json.dumps({"plain_ms": round(t_plain, 3), "graph_ms": round(t_graph, 3), "rel_err": err}))
