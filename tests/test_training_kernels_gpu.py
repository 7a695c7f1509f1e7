"""The small training-side entry points of the C ABI, each against the tensor expression it replaces (plain torch on the
same device tensors, float64 where a sum is involved).  The gradient tests of test_fno_gpu.py / test_ns2d_gpu.py exercise
them inside the autograd nodes; here they are called directly, with shapes that do not divide evenly."""
import ctypes

import pytest
import torch

from conftest import rel_l2

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def dev():
    return torch.device("cuda:0")


@pytest.fixture(scope="module")
def lib():
    from torch_cfd_amd import _lib

    return _lib.load()


def _stream(dev):
    return ctypes.c_void_p(torch.cuda.current_stream(dev).cuda_stream)


@pytest.mark.parametrize("rows,cols", [(2048, 1816), (37, 5), (1, 300), (4100, 33)])
def test_sum_rows(rows, cols, dev, lib):
    from torch_cfd_amd import _lib

    g = torch.Generator().manual_seed(rows)
    m = torch.randn(rows, cols, generator=g).to(dev)
    out = torch.empty(cols, dtype=torch.float64, device=dev)
    scratch = torch.empty(lib.tcfd_sum_rows_slices(rows) * cols, dtype=torch.float64, device=dev)
    _lib.check(lib.tcfd_sum_rows(m.data_ptr(), out.data_ptr(), scratch.data_ptr(), rows, cols, _stream(dev)), "tcfd_sum_rows")
    assert rel_l2(out, m.double().sum(0)) < 1e-14


@pytest.mark.parametrize("rows,T,sT", [(1000, 10, 10), (77, 7, 3), (64, 4, 1)])
def test_sum_t_into_last(rows, T, sT, dev, lib):
    from torch_cfd_amd import _lib

    d = torch.randn(rows, T, generator=torch.Generator().manual_seed(T)).to(dev)
    g = torch.full((rows, sT), 7.0, device=dev)
    _lib.check(lib.tcfd_sum_t_into_last(d.data_ptr(), g.data_ptr(), rows, T, sT, _stream(dev)), "tcfd_sum_t_into_last")
    ref = torch.zeros(rows, sT, device=dev)
    ref[:, -1] = d.sum(-1)
    assert torch.allclose(g, ref, rtol=1e-6, atol=1e-6) and float(g[:, :-1].abs().max() if sT > 1 else 0.0) == 0.0


@pytest.mark.parametrize("C,co", [(7, 5), (15, 16), (16, 16), (20, 20), (32, 32), (31, 17), (40, 9)])
@pytest.mark.parametrize("with_table", [False, True])
def test_sample_outer_sums(with_table, C, co, dev, lib):
    from torch_cfd_amd import _lib

    g = torch.Generator().manual_seed(3)
    b, P, wps = 3, 16 * 41, 8
    R16, C16 = 16 * ((co + 15) // 16), 16 * ((C + 16) // 16)          # wide layers: more 16 x 16 tiles (C + 1 columns, co rows)
    dy = torch.randn(b, co, P, generator=g).to(dev)
    if with_table:
        x1, pe = torch.randn(b, P, generator=g).to(dev), torch.randn(C, P, generator=g).to(dev)
        xin, xp, pp = x1[:, None] + pe[None], x1.data_ptr(), pe.data_ptr()
    else:
        xin = torch.randn(b, C, P, generator=g).to(dev)
        xp, pp = xin.data_ptr(), None
    tiles = torch.empty(wps, b, R16 * C16, device=dev)
    _lib.check(lib.tcfd_fno_sample_outer_sums(dy.data_ptr(), xp, pp, tiles.data_ptr(), b, C, co, P, wps, _stream(dev)),
               "tcfd_fno_sample_outer_sums")
    M = tiles.double().sum(0).view(b, R16, C16)
    assert rel_l2(M[:, :co, :C], torch.einsum("bop,bcp->boc", dy.double(), xin.double())) < 1e-6
    assert rel_l2(M[:, :co, C], dy.double().sum(-1)) < 1e-6
    assert float(M[:, co:].abs().max() if co < R16 else 0.0) == 0.0 and float(M[:, :, C + 1:].abs().max() if C + 1 < C16 else 0.0) == 0.0
    # loud on shapes it does not cover
    assert lib.tcfd_fno_sample_outer_sums(dy.data_ptr(), xp, pp, tiles.data_ptr(), b, C, co, P - 4, wps, _stream(dev)) != 0


@pytest.mark.parametrize("cdtype", [torch.complex128, torch.complex64])
def test_stage_update_and_its_vjp(cdtype, dev, lib):
    from torch_cfd_amd import _lib

    real = torch.float64 if cdtype == torch.complex128 else torch.float32
    code = _lib.TCFD_C128 if cdtype == torch.complex128 else _lib.TCFD_C64
    g = torch.Generator().manual_seed(9)
    B, n, m = 3, 12, 7
    rnd = lambda *s: torch.complex(torch.randn(*s, generator=g, dtype=real), torch.randn(*s, generator=g, dtype=real)).to(dev)
    f, hp, bb, gu, gh = (rnd(B, n, m) for _ in range(5))
    lin = (-torch.rand(n, m, generator=g, dtype=real)).to(dev)
    fa, beta, gdt, mu, mud = 0.7, -0.4, 1e-2, 3e-3, 5e-3
    coef = (ctypes.c_double * 5)(fa, beta, gdt, mu, mud)
    h, u = torch.empty_like(f), torch.empty_like(f)
    tol = 1e-14 if real == torch.float64 else 2e-6
    for prev in (hp, None):
        _lib.check(lib.tcfd_ns2d_stage_update(f.data_ptr(), prev.data_ptr() if prev is not None else None, bb.data_ptr(),
                                              lin.data_ptr(), coef, h.data_ptr(), u.data_ptr(), B, n * m, code, _stream(dev)),
                   "tcfd_ns2d_stage_update")
        h_ref = fa * f + (beta * prev if prev is not None else 0)
        u_ref = (bb + gdt * h_ref + mu * lin * bb) / (1 - mud * lin)
        assert rel_l2(h, h_ref) < tol and rel_l2(u, u_ref) < tol
    gf, ghp, gb = torch.empty_like(f), torch.empty_like(f), torch.empty_like(f)
    _lib.check(lib.tcfd_ns2d_stage_update_vjp(gu.data_ptr(), gh.data_ptr(), lin.data_ptr(), coef, gf.data_ptr(), ghp.data_ptr(),
                                              gb.data_ptr(), B, n * m, code, _stream(dev)), "tcfd_ns2d_stage_update_vjp")
    r = 1 / (1 - mud * lin)
    G = gh + gdt * r * gu
    assert rel_l2(gf, fa * G) < tol and rel_l2(ghp, beta * G) < tol and rel_l2(gb, (1 + mu * lin) * r * gu) < tol


def test_vjp_combine(dev, lib):
    from torch_cfd_amd import _lib

    g = torch.Generator().manual_seed(1)
    B, n, m = 3, 10, 6
    rnd = lambda *s: torch.complex(torch.randn(*s, generator=g, dtype=torch.float64), torch.randn(*s, generator=g, dtype=torch.float64)).to(dev)
    X, post = rnd(4, B, n, m), rnd(4, n, m)
    out = torch.empty(B, n, m, dtype=torch.complex128, device=dev)
    _lib.check(lib.tcfd_ns2d_vjp_combine(X.data_ptr(), post.data_ptr(), out.data_ptr(), B, n * m, _lib.TCFD_C128, _stream(dev)),
               "tcfd_ns2d_vjp_combine")
    assert rel_l2(out, (X * post[:, None]).sum(0)) < 1e-14


def test_inverse_transform_adds_to_an_existing_tensor(dev):
    """tcfd_fno_inverse_trunc_acc through hip_truncated_irfftn(accumulate=...): out = acc + transform, in place, odd t_keep too."""
    from torch_cfd_amd import fno

    g = torch.Generator().manual_seed(4)
    b, c, X, Y, T, modes = 2, 3, 32, 32, 8, (4, 6, 3)
    for t_keep in (8, 6):
        plan = fno._plan((X, Y, T, 0, T) + modes, dev, torch.float32)
        vh = torch.view_as_complex(torch.randn(b, c, 2 * modes[0], 2 * modes[1], modes[2], 2, generator=g)).to(dev)
        plain = fno.hip_truncated_irfftn(vh, plan, t_keep)
        acc = torch.randn(b, c, X, Y, t_keep, generator=g).to(dev)
        before = acc.clone()
        out = fno.hip_truncated_irfftn(vh, plan, t_keep, accumulate=acc)
        assert out.data_ptr() == acc.data_ptr() and torch.allclose(out, before + plain, rtol=1e-6, atol=1e-6)
    with pytest.raises(ValueError):
        fno.hip_truncated_irfftn(vh, plan, 6, accumulate=torch.zeros(b, c, X, Y, 8, device=dev))
