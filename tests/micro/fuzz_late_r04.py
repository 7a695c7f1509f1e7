"""Random configurations through the late-round-4 paths (lifting operator through the spectrum, 71-product backward, compact skip
gradient, chunked layers): every switch on against every switch off, forward and gradients.  python tests/micro/fuzz_late_r04.py [n]"""
import itertools, os, random, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from torch_cfd_amd import fno

dev = torch.device("cuda:0")
rel = lambda a, b: float((a.double() - b.double()).norm() / b.double().norm().clamp_min(1e-30))
SW = ("TCFD_LIFT_SPECTRUM", "TCFD_PW_BWD_YMASK", "TCFD_COMPACT_SKIP_GRAD", "TCFD_FNO_CHUNK_MB")
ON = {"TCFD_LIFT_SPECTRUM": "1", "TCFD_PW_BWD_YMASK": "2", "TCFD_COMPACT_SKIP_GRAD": "1", "TCFD_FNO_CHUNK_MB": "0.3"}
OFF = {"TCFD_LIFT_SPECTRUM": "0", "TCFD_PW_BWD_YMASK": "0", "TCFD_COMPACT_SKIP_GRAD": "0", "TCFD_FNO_CHUNK_MB": "0"}
rng = random.Random(0)
n = int(sys.argv[1]) if len(sys.argv) > 1 else 24
worst = {"fwd": 0.0, "grad": 0.0}
for it in range(n):
    width = rng.choice([4, 6, 8, 10, 12])
    grid = rng.choice([16, 24, 32, 48, 80, 96])
    T = rng.choice([6, 10])
    mx = rng.choice([2, 4]) if grid < 32 else rng.choice([4, 8])
    mt = rng.choice([2, 3])
    act = rng.choice(["ReLU", "GELU"])
    kw = dict(width=width, num_spectral_layers=rng.choice([1, 2]), latent_steps=T, activation=act,
              temporal_padding=rng.choice([True, False]), spatial_random_feats=rng.choice([False, True]))
    b = rng.choice([1, 3, 4])
    torch.manual_seed(it)
    try:
        model = fno.SFNO(mx, mx, mt, **kw).to(dev)
    except Exception as e:
        print("config rejected at construction:", kw, repr(e)[:80]); continue
    x = torch.randn(b, grid, grid, T, device=dev)
    y = torch.randn(b, grid, grid, T, device=dev)
    loss_fn = fno.SobolevLoss(n_grid=grid, norm_order=0, relative=True).to(dev)
    res = {}
    for name, env in (("on", ON), ("off", OFF)):
        os.environ.update(env)
        model.eval()
        with torch.no_grad():
            fwd = model(x)
        model.train()
        model.zero_grad(set_to_none=True)
        loss_fn(model(x), y).backward()
        res[name] = (fwd, [p.grad.clone() for p in model.parameters() if p.grad is not None])
    ef = rel(res["on"][0], res["off"][0])
    eg = max(rel(a, c) for a, c in zip(res["on"][1], res["off"][1]) if float(c.abs().max()) > 0)
    worst["fwd"], worst["grad"] = max(worst["fwd"], ef), max(worst["grad"], eg)
    flag = "" if (ef < 1e-5 and eg < 1e-4 and torch.isfinite(res["on"][0]).all()) else "   <-- LOOK"
    print(f"{it:3d} w{width:2d} n{grid:3d} T{T:2d} m{mx}/{mt} {act:4s} b{b} pad{int(kw['temporal_padding'])} rf{int(kw['spatial_random_feats'])} "
          f"L{kw['num_spectral_layers']}: fwd {ef:.1e} grad {eg:.1e}{flag}")
print("worst", worst)
