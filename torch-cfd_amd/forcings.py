"""State-independent body forces of the spectral solver.

Same constructor arguments and call protocol as the reference forcings
(torch_cfd/forcings.py:61-349): ``fn(grid, field)`` returns physical-space
arrays exposing ``.data``; ``fn.vorticity`` says whether it is a force on the
vorticity (one array) or on the velocity (an (fx, fy) pair whose curl is taken,
torch_cfd/equations.py:429-437).

The reference re-evaluates the forcing (meshgrid + sin + 2 rfft2) inside every
RK stage although none of its forcings depends on the state; here the operator
evaluates it ONCE, transforms it, and hands the (n, m) complex table to the HIP
plan (include/tcfd.h: ``forcing_hat``).
"""
from __future__ import annotations

from typing import Optional

import torch
import torch.nn as nn

from .grids import Grid


class FieldArray:
    """Minimal stand-in for the reference's GridArray: data + offset + grid."""

    def __init__(self, data: torch.Tensor, offset, grid: Grid):
        self.data = data
        self.offset = tuple(offset)
        self.grid = grid

    @property
    def shape(self):
        return self.data.shape


class ForcingFn(nn.Module):
    def __init__(self, grid: Grid, scale: float = 1, wave_number: int = 1, diam: float = 1.0,
                 swap_xy: bool = False, vorticity: bool = False, offsets=None, device=None, **kwargs):
        super().__init__()
        self.grid = grid
        self.scale = scale
        self.wave_number = wave_number
        self.diam = diam
        self.swap_xy = swap_xy
        self.vorticity = vorticity
        self.offsets = grid.cell_faces if offsets is None else offsets
        self.device = grid.device if device is None else device

    def velocity_eval(self, grid, velocity=None):
        raise NotImplementedError

    def vorticity_eval(self, grid, vorticity=None):
        raise NotImplementedError

    def forward(self, grid: Optional[Grid] = None, velocity=None, vorticity=None):
        if not self.vorticity:
            return self.velocity_eval(grid, velocity)
        return self.vorticity_eval(grid, vorticity)


class KolmogorovForcing(ForcingFn):
    """fx = scale * sin(k * y) (or fy = scale * sin(k * x) with swap_xy), k = wave_number * 2 pi / diam;
    vorticity form: -scale * k * cos(k * y).  (torch_cfd/forcings.py:118-210)"""

    def __init__(self, diam=2 * torch.pi, offsets=((0, 0), (0, 0)), vorticity=False, *args, **kwargs):
        super().__init__(*args, diam=diam, offsets=offsets, vorticity=vorticity, **kwargs)

    def velocity_eval(self, grid, velocity=None):
        grid = self.grid if grid is None else grid
        k = self.wave_number * (2 * torch.pi / self.diam)
        if self.swap_xy:
            x = grid.mesh(self.offsets[1])[0]
            v = FieldArray(self.scale * torch.sin(k * x), self.offsets[1], grid)
            u = FieldArray(torch.zeros_like(v.data), (1, 1 / 2), grid)
        else:
            y = grid.mesh(self.offsets[0])[1]
            u = FieldArray(self.scale * torch.sin(k * y), self.offsets[0], grid)
            v = FieldArray(torch.zeros_like(u.data), (1 / 2, 1), grid)
        return u, v

    def vorticity_eval(self, grid, vorticity=None):
        grid = self.grid if grid is None else grid
        k = self.wave_number * (2 * torch.pi / self.diam)
        if self.swap_xy:
            s, off = grid.mesh(self.offsets[1])[0], self.offsets[1]
        else:
            s, off = grid.mesh(self.offsets[0])[1], self.offsets[0]
        return FieldArray(-self.scale * k * torch.cos(k * s), off, grid)


class SimpleSolenoidalForcing(ForcingFn):
    """Divergence-free forcing F = (psi, -psi) given by a scalar potential
    (torch_cfd/forcings.py:220-302); subclasses provide the potentials."""

    def __init__(self, scale=1, diam=1.0, k=1.0, offsets=((0, 0), (0, 0)), vorticity=True, *args, **kwargs):
        super().__init__(*args, scale=scale, diam=diam, wave_number=k, offsets=offsets,
                         vorticity=vorticity, **kwargs)

    def potential(self, x, y, s, k):
        raise NotImplementedError

    def vort_potential(self, x, y, s, k):
        raise NotImplementedError

    def velocity_eval(self, grid, velocity=None):
        grid = self.grid if grid is None else grid
        k = self.wave_number * (2 * torch.pi / self.diam)
        s = 0.5 * self.scale / (2 * torch.pi) / self.wave_number
        if self.swap_xy:
            x, y = grid.mesh(self.offsets[1])[0], grid.mesh(self.offsets[0])[1]
            rot = self.potential(x, y, s, k)
            return FieldArray(-rot, (1, 1 / 2), grid), FieldArray(rot, self.offsets[1], grid)
        x, y = grid.mesh(self.offsets[0])[0], grid.mesh(self.offsets[1])[1]
        rot = self.potential(x, y, s, k)
        return FieldArray(rot, self.offsets[0], grid), FieldArray(-rot, (1 / 2, 1), grid)

    def vorticity_eval(self, grid, vorticity=None):
        grid = self.grid if grid is None else grid
        k = self.wave_number * (2 * torch.pi / self.diam)
        if self.swap_xy:
            x, y = grid.mesh(self.offsets[1])[0], grid.mesh(self.offsets[0])[1]
        else:
            x, y = grid.mesh(self.offsets[0])[0], grid.mesh(self.offsets[1])[1]
        return self.vort_potential(x, y, self.scale, k)


class SinCosForcing(SimpleSolenoidalForcing):
    """The FNO-paper forcing a*(sin(k(x+y)) + cos(k(x+y))) on the vorticity
    (torch_cfd/forcings.py:305-349)."""

    def __init__(self, scale=0.1, diam=1.0, k=1.0, offsets=((0, 0), (0, 0)), *args, **kwargs):
        super().__init__(*args, scale=scale, diam=diam, k=k, offsets=offsets, **kwargs)

    def potential(self, x, y, s, k):
        return s * (torch.sin(k * (x + y)) - torch.cos(k * (x + y)))

    def vort_potential(self, x, y, s, k):
        return s * (torch.cos(k * (x + y)) + torch.sin(k * (x + y)))
