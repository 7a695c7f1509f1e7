"""Body forces of the spectral solver.

A forcing of this path is *state independent*: the operator samples it ONCE when it builds its device plan,
transforms it, and hands the (n, m) spectrum to the kernels as a table (include/tcfd.h ``forcing_hat``) -- the
reference re-evaluates mesh + sin + two rfft2 inside every RK stage (torch_cfd/equations.py:429-437).

Call protocol (what ``NavierStokes2DSpectral.forcing_hat`` consumes, and what a user-defined forcing has to offer):
``fn(grid, field)`` returns sampled physical-space data -- objects with a ``.data`` tensor -- either ONE array when
``fn.vorticity`` is true (a force on the vorticity equation) or an ``(fx, fy)`` pair (a force on the momentum
equation; the operator takes its curl).  Constructor keywords follow the reference classes
(torch_cfd/forcings.py:118-210 Kolmogorov, :305-349 sin/cos) so that scripts carry over.
"""
from __future__ import annotations

import math
from typing import Optional, Sequence, Tuple

import torch
import torch.nn as nn

from .grids import Grid

_CORNERS = ((0, 0), (0, 0))   # both components sampled at the cell corners x_i = lo + i h (the reference's default)


class FieldArray:
    """Sampled field: ``data`` plus where it was sampled (``offset`` in cells, ``grid``)."""

    __slots__ = ("data", "offset", "grid")

    def __init__(self, data: torch.Tensor, offset: Optional[Sequence[float]] = None, grid: Optional[Grid] = None):
        self.data = data
        self.offset = None if offset is None else tuple(offset)
        self.grid = grid

    @property
    def shape(self):
        return self.data.shape


class ForcingFn(nn.Module):
    """Base of the built-in forcings: a plane wave family with angular wavenumber ``wave_number * 2 pi / diam``.

    Subclasses implement ``momentum(x, y)`` -> (fx, fy) and / or ``curl(x, y)`` -> the same force acting on the
    vorticity equation, both on coordinate arrays.  ``fingerprint()`` identifies the sampled table (the operator
    keys its device plan on it, so changing e.g. ``scale`` after construction rebuilds the plan)."""

    def __init__(self, grid: Grid, scale: float = 1.0, wave_number: float = 1, diam: float = 1.0, swap_xy: bool = False,
                 vorticity: bool = False, offsets=None, device=None):
        super().__init__()
        self.grid = grid
        self.scale = scale
        self.wave_number = wave_number
        self.diam = diam
        self.swap_xy = swap_xy
        self.vorticity = vorticity
        self.offsets = _CORNERS if offsets is None else tuple(tuple(o) for o in offsets)
        self.device = grid.device if device is None else device

    @property
    def angular_wavenumber(self) -> float:
        return self.wave_number * (2 * math.pi / self.diam)

    def fingerprint(self) -> tuple:
        return (type(self).__qualname__, float(self.scale), float(self.wave_number), float(self.diam), bool(self.swap_xy),
                bool(self.vorticity), self.offsets)

    def momentum(self, x: torch.Tensor, y: torch.Tensor) -> Tuple[torch.Tensor, torch.Tensor]:
        raise NotImplementedError(f"{type(self).__name__} has no momentum-equation form")

    def curl(self, x: torch.Tensor, y: torch.Tensor) -> torch.Tensor:
        raise NotImplementedError(f"{type(self).__name__} has no vorticity-equation form")

    # The reference's extension points (torch_cfd/forcings.py:97-116): a user forcing written against the reference
    # overrides ``velocity_eval(grid, velocity)`` and / or ``vorticity_eval(grid, vorticity)`` and returns sampled
    # data; ``forward`` dispatches on ``self.vorticity`` exactly as there.  The built-in family implements the two
    # hooks once, here, on top of ``momentum`` / ``curl``.
    def velocity_eval(self, grid: Optional[Grid], velocity=None):
        grid = self.grid if grid is None else grid
        off_x, off_y = self.offsets
        (x, _), (_, y) = grid.mesh(off_x), grid.mesh(off_y)
        fx, fy = self.momentum(x, y)
        return FieldArray(fx, off_x, grid), FieldArray(fy, off_y, grid)

    def vorticity_eval(self, grid: Optional[Grid], vorticity=None):
        grid = self.grid if grid is None else grid
        off_x = self.offsets[0]
        x, y = grid.mesh(off_x)
        return FieldArray(self.curl(x, y), off_x, grid)

    def forward(self, grid: Optional[Grid] = None, velocity=None, vorticity=None):
        if self.vorticity:
            return self.vorticity_eval(grid, vorticity)
        return self.velocity_eval(grid, velocity)


class KolmogorovForcing(ForcingFn):
    """Shear forcing ``f = scale * sin(k y) e_x`` (``swap_xy``: ``scale * sin(k x) e_y``); on the vorticity equation the
    reference applies ``-scale * k * cos(k s)`` along the sheared coordinate s in both cases."""

    def __init__(self, grid: Grid, scale: float = 1.0, wave_number: float = 1, diam: float = 2 * math.pi,
                 swap_xy: bool = False, vorticity: bool = False, offsets=None, device=None, **unused):
        super().__init__(grid, scale=scale, wave_number=wave_number, diam=diam, swap_xy=swap_xy, vorticity=vorticity,
                         offsets=offsets, device=device)

    def momentum(self, x, y):
        k = self.angular_wavenumber
        if self.swap_xy:
            wave = self.scale * torch.sin(k * x)
            return torch.zeros_like(wave), wave
        wave = self.scale * torch.sin(k * y)
        return wave, torch.zeros_like(wave)

    def curl(self, x, y):
        k = self.angular_wavenumber
        s = x if self.swap_xy else y
        return -self.scale * k * torch.cos(k * s)


class SimpleSolenoidalForcing(ForcingFn):
    """Template of a divergence-free forcing given by ONE scalar profile (torch_cfd/forcings.py:220-297): the momentum
    form is the pair ``(g, -g)`` with ``g = potential(x, y, a, k)``, ``a = scale / (4 pi wave_number)``, and the force
    on the vorticity equation is ``vort_potential(x, y, scale, k)``, ``k`` the angular wavenumber.  ``swap_xy`` flips
    the sign of the pair.  Subclasses supply the two profiles (plain functions of ``(x, y, amplitude, k)``)."""

    def __init__(self, grid: Grid, scale: float = 1, diam: float = 1.0, k: float = 1.0, swap_xy: bool = False,
                 vorticity: bool = True, offsets=None, device=None, **unused):
        super().__init__(grid, scale=scale, wave_number=k, diam=diam, swap_xy=swap_xy, vorticity=vorticity,
                         offsets=offsets, device=device)

    def potential(self, x, y, s, k):
        raise NotImplementedError(f"{type(self).__name__}.potential")

    def vort_potential(self, x, y, s, k):
        raise NotImplementedError(f"{type(self).__name__}.vort_potential")

    def momentum(self, x, y):
        g = self.potential(x, y, self.scale / (4 * math.pi * self.wave_number), self.angular_wavenumber)
        return (-g, g) if self.swap_xy else (g, -g)

    def curl(self, x, y):
        return self.vort_potential(x, y, self.scale, self.angular_wavenumber)


class SinCosForcing(SimpleSolenoidalForcing):
    """The FNO-paper forcing ``scale * (sin(k (x + y)) + cos(k (x + y)))`` on the vorticity equation (the default
    here as in the reference, ``vorticity=True``); momentum profile ``a (sin(k (x + y)) - cos(k (x + y)))``."""

    def __init__(self, grid: Grid, scale: float = 0.1, diam: float = 1.0, k: float = 1.0, swap_xy: bool = False,
                 vorticity: bool = True, offsets=None, device=None, **unused):
        super().__init__(grid, scale=scale, diam=diam, k=k, swap_xy=swap_xy, vorticity=vorticity, offsets=offsets,
                         device=device)

    def potential(self, x, y, s, k):
        phase = k * (x + y)
        return s * (torch.sin(phase) - torch.cos(phase))

    def vort_potential(self, x, y, s, k):
        phase = k * (x + y)
        return s * (torch.cos(phase) + torch.sin(phase))
