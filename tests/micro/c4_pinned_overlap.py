"""Where does the one-batch C4 job (rank 0 of 8) lose its time?  (a) page-locking the 5.4 GB result alone; (b) the 64-sample job
with a 64-sample result (what a peer rank costs); (c) the same with the full 512-sample result allocated on the helper thread;
(d) stepping only while a helper thread page-locks / first-touches memory (does the allocation stall kernel launches?)."""
import json, os, sys, threading, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from torch_cfd_amd.data_gen import generate_mcwilliams_dataset
import torch_cfd_amd as tc
dev = torch.device("cuda", 0)
torch.set_default_dtype(torch.float64)
res = {}
def job(total, as_rank0_of=None):
    st = {}
    torch.cuda.synchronize(); t0 = time.perf_counter()
    generate_mcwilliams_dataset(512, total, 64, 1e-3, 100, 550, 55, viscosity=1e-3, peak_wavenumber=4, random_state=0, subsample=2,
                                dtype=torch.float32, cdtype=torch.complex64, device=dev, stats=st, as_rank0_of=as_rank0_of)
    torch.cuda.synchronize()
    st["seconds"] = time.perf_counter() - t0
    return {k: round(v, 3) if isinstance(v, float) else v for k, v in st.items()}
job(64)                                     # warm-up: kernels loaded, allocator warm
torch._C._host_emptyCache()
res["b_64_samples_small_result"] = job(64)
torch._C._host_emptyCache()
res["b2_64_samples_small_result_again"] = job(64)
torch._C._host_emptyCache()
t0 = time.perf_counter()
bufs = [torch.empty((512, 10, 256, 256), dtype=torch.float32, pin_memory=True) for _ in range(4)]
res["a_pin_5p4GB_alone_s"] = round(time.perf_counter() - t0, 3)
del bufs; torch._C._host_emptyCache()
res["c_rank0_of_8_full_result_on_helper_thread"] = job(512, as_rank0_of=8)
torch._C._host_emptyCache()
# (d) raw stepping under a helper thread
import math
L = 2 * math.pi
grid = tc.Grid(shape=(512, 512), domain=((0, L), (0, L)), device=dev)
op = tc.NavierStokes2DSpectral(1e-3, grid, drag=0, smooth=True, solver=tc.RK4CrankNicolsonStepper()).to(dev)
from torch_cfd_amd.initial_conditions import vorticity_field
with torch.no_grad():
    w = tc.fft_plan(512, torch.complex128, dev).rfft2(vorticity_field(grid, 4, batch_seeds=list(range(64)), device=dev))
    def stepping(k=300):
        global w
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(k // 50): w, _ = op(w, 1e-3, steps=50)
        torch.cuda.synchronize(); return time.perf_counter() - t0
    stepping(100)
    res["d0_300_steps_alone_s"] = round(stepping(), 3)
    def helper(kind):
        t0 = time.perf_counter()
        if kind == "pin":
            hb = [torch.empty((512, 10, 256, 256), dtype=torch.float32, pin_memory=True) for _ in range(4)]
        elif kind == "pin_small":
            hb = [torch.empty((8, 10, 256, 256), dtype=torch.float32, pin_memory=True) for _ in range(256)]
        else:
            hb = [torch.empty((512, 10, 256, 256), dtype=torch.float32).fill_(0) for _ in range(4)]
        helper.t = time.perf_counter() - t0
        helper.keep = hb
    for kind in ("pin", "pin_small", "touch"):
        th = threading.Thread(target=helper, args=(kind,)); th.start()
        res[f"d_{kind}_300_steps_s"] = round(stepping(), 3)
        th.join(); res[f"d_{kind}_helper_s"] = round(helper.t, 3)
        helper.keep = None; torch._C._host_emptyCache()
print(json.dumps(res))
