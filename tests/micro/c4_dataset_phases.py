import sys, time, torch
sys.path.insert(0, '/root/repo')
from torch_cfd_amd.data_gen import generate_mcwilliams_dataset
import torch_cfd_amd.data_gen as dg
dev = torch.device('cuda', 0)
torch.set_default_dtype(torch.float64)
for rep in range(3):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    data = generate_mcwilliams_dataset(512, 64, 64, 1e-3, 100, 550, 55, viscosity=1e-3, peak_wavenumber=4, random_state=rep,
                                       subsample=2, dtype=torch.float32, cdtype=torch.complex64, device=dev, path=None)
    torch.cuda.synchronize(); print('run', rep, round(time.perf_counter() - t0, 3), 's')
import cProfile, pstats
pr = cProfile.Profile(); pr.enable()
data = generate_mcwilliams_dataset(512, 64, 64, 1e-3, 100, 550, 55, viscosity=1e-3, peak_wavenumber=4, random_state=7,
                                   subsample=2, dtype=torch.float32, cdtype=torch.complex64, device=dev, path=None)
torch.cuda.synchronize(); pr.disable()
pstats.Stats(pr).sort_stats('cumulative').print_stats(14)
