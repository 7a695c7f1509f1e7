import json, os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
os.environ["TCFD_HANDOVER_TRACE"] = "1"
from torch_cfd_amd.data_gen import generate_mcwilliams_dataset
dev = torch.device("cuda", 0)
torch.set_default_dtype(torch.float64)
def job(total, as_rank0_of=None):
    st = {}
    torch.cuda.synchronize(); t0 = time.perf_counter()
    generate_mcwilliams_dataset(512, total, 64, 1e-3, 100, 550, 55, viscosity=1e-3, peak_wavenumber=4, random_state=0, subsample=2,
                                dtype=torch.float32, cdtype=torch.complex64, device=dev, stats=st, as_rank0_of=as_rank0_of)
    torch.cuda.synchronize(); st["seconds"] = round(time.perf_counter() - t0, 3)
    return st
job(64); torch._C._host_emptyCache()
a = job(64); torch._C._host_emptyCache()
b = job(512, as_rank0_of=8)
for name, st in (("64 alone", a), ("rank 0 of 8", b)):
    print(name, {k: (round(v, 3) if isinstance(v, float) else v) for k, v in st.items() if k != "trace"})
    print("   ", st["trace"])
