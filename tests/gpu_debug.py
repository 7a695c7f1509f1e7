"""One-shot diagnostic sweep for a gpurun call (not a pytest file): prints the
rel-L2 error of every HIP entry point against the CPU oracle over sizes/dtypes,
so a single GPU round trip localises a bug."""
import math
import os
import sys
import time
import traceback

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch_cfd_amd as tc  # noqa: E402
from oracle import ns2d as O  # noqa: E402

L = 2 * math.pi
dev = torch.device("cuda:0")


def rel(a, b):
    a = a.detach().cpu()
    b = b.detach().cpu()
    if a.is_complex():
        a, b = a.to(torch.complex128), b.to(torch.complex128)
    else:
        a, b = a.double(), b.double()
    return (torch.linalg.norm((a - b).reshape(-1)) / torch.linalg.norm(b.reshape(-1))).item()


def run(label, fn):
    try:
        t0 = time.time()
        msg = fn()
        torch.cuda.synchronize()
        print(f"[{label}] {msg}  ({time.time()-t0:.2f}s)", flush=True)
    except Exception:
        print(f"[{label}] EXCEPTION\n{traceback.format_exc()}", flush=True)


def main():
    sizes = [int(s) for s in os.environ.get("SIZES", "8,16,32,64,128,256,512,1024,2048").split(",")]
    print(torch.cuda.get_device_name(0), torch.version.hip, flush=True)
    for real, cdt in ((torch.float64, torch.complex128), (torch.float32, torch.complex64)):
        torch.set_default_dtype(real)
        for n in sizes:
            B = 3 if n <= 256 else (2 if n <= 1024 else 1)
            m = n // 2 + 1
            g = torch.Generator().manual_seed(n)
            plan = tc.fft_plan(n, cdt, dev)

            def t_rfft2():
                x = torch.randn(B, n, n, generator=g, dtype=real)
                return f"rfft2 err={rel(plan.rfft2(x.to(dev)), torch.fft.rfft2(x)):.2e}"

            def t_irfft2():
                xh = torch.view_as_complex(torch.randn(B, n, m, 2, generator=g, dtype=real))
                return f"irfft2(non-hermitian) err={rel(plan.irfft2(xh.to(dev)), torch.fft.irfft2(xh)):.2e}"

            run(f"{cdt} n={n}", t_rfft2)
            run(f"{cdt} n={n}", t_irfft2)

            grid = tc.Grid(shape=(n, n), domain=((0, L), (0, L)))
            forcing = tc.KolmogorovForcing(grid=grid, scale=1.0, wave_number=4) if n >= 16 else None
            op = tc.NavierStokes2DSpectral(1e-3, grid, drag=0.1, forcing_fn=forcing,
                                           solver=tc.RK4CrankNicolsonStepper()).to(dev)
            t = O.make_tables(n, L, 1e-3, 0.1, True, None, real)
            if forcing is not None:
                t.forcing_hat = O.kolmogorov_forcing_hat(n, L, t.kx, t.ky, 1.0, 4, real=real)
            w0 = torch.stack([torch.fft.rfft2(O.mcwilliams_vorticity(n, L, 4, s, real)) for s in range(B)])
            wd = w0.to(dev)

            def t_vel():
                (uh, vh), psi = tc.vorticity_to_velocity(grid, wd, (op.kx, op.ky))
                (ruh, rvh), rpsi = O.stream_and_velocity(w0, t.kx, t.ky)
                return f"velocity err u={rel(uh, ruh):.2e} v={rel(vh, rvh):.2e} psi={rel(psi, rpsi):.2e}"

            def t_F():
                return f"F err={rel(op.explicit_terms(wd), O.explicit_terms(w0, t)):.2e}"

            def t_step():
                o1, d1 = op(wd, 1e-3)
                r1, rd1 = O.advance(w0, 1e-3, t)
                res = op.residual(o1, d1)
                rres = O.residual(r1, rd1, t)
                return f"step1 err w={rel(o1, r1):.2e} dwdt={rel(d1, rd1):.2e} residual={rel(res, rres):.2e}"

            def t_step5():
                o, d = op(wd, 1e-3, steps=5)
                r, rd = O.advance(w0, 1e-3, t, steps=5)
                return f"step5 err w={rel(o, r):.2e} dwdt={rel(d, rd):.2e}"

            run(f"{cdt} n={n}", t_vel)
            run(f"{cdt} n={n}", t_F)
            run(f"{cdt} n={n}", t_step)
            if n <= 512:
                run(f"{cdt} n={n}", t_step5)
    torch.set_default_dtype(torch.float32)


if __name__ == "__main__":
    main()
