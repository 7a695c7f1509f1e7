"""Periodic-box grid descriptor: the constructor argument of the spectral operator.

Mirrors the part of the reference ``Grid`` the spectral path touches
(torch_cfd/grids.py:37-218: shape/step/domain, ``axes``, ``mesh``, ``fft_axes``,
``fft_mesh``, ``rfft_mesh``).  The staggered-grid data model of the reference
(GridArray/GridVariable, grids.py:221-1095) belongs to its finite-volume solver
and is out of scope.
"""
from __future__ import annotations

import math
import numbers
import operator
from typing import Optional, Sequence, Tuple, Union

import torch


class Grid:
    def __init__(
        self,
        shape: Sequence[int],
        step: Optional[Union[float, Sequence[float]]] = None,
        domain: Optional[Union[float, Sequence[Tuple[float, float]]]] = None,
        device: Optional[Union[str, torch.device]] = "cpu",
    ):
        shape = tuple(operator.index(s) for s in shape)
        ndim = len(shape)
        if step is not None and domain is not None:
            raise TypeError("cannot provide both step and domain")
        if domain is not None:
            if isinstance(domain, (int, float)):
                domain = ((0, domain),) * ndim
            elif len(domain) != ndim:
                raise ValueError(f"length of domain does not match ndim: {len(domain)} != {ndim}")
            for bounds in domain:
                if len(bounds) != 2:
                    raise ValueError(f"domain is not sequence of pairs of numbers: {domain}")
            domain = tuple((float(lo), float(hi)) for lo, hi in domain)
        else:
            if step is None:
                step = 1
            if isinstance(step, numbers.Number):
                step = (step,) * ndim
            elif len(step) != ndim:
                raise ValueError(f"length of step does not match ndim: {len(step)} != {ndim}")
            domain = tuple((0.0, float(s * n)) for s, n in zip(step, shape))
        self.shape = shape
        self.domain = domain
        self.step = tuple((hi - lo) / n for (lo, hi), n in zip(domain, shape))
        self.device = device

    @property
    def ndim(self) -> int:
        return len(self.shape)

    @property
    def cell_center(self) -> Tuple[float, ...]:
        return self.ndim * (0.5,)

    @property
    def cell_faces(self):
        d = self.ndim
        return tuple(tuple(1.0 if i == j else 0.5 for j in range(d)) for i in range(d))

    def axes(self, offset: Optional[Sequence[float]] = None):
        """Grid points along each axis, shifted by ``offset * step`` (default: cell centres)."""
        if offset is None:
            offset = self.cell_center
        if len(offset) != self.ndim:
            raise ValueError(f"unexpected offset length: {len(offset)} vs {self.ndim}")
        return tuple(
            lo + (torch.arange(n) + off) * h
            for (lo, _), off, n, h in zip(self.domain, offset, self.shape, self.step)
        )

    def mesh(self, offset: Optional[Sequence[float]] = None):
        x, y = torch.meshgrid(*self.axes(offset), indexing="ij")
        return x.to(self.device), y.to(self.device)

    def fft_axes(self):
        """Ordinal frequencies per axis (multiply by 2*pi for angular ones)."""
        return tuple(torch.fft.fftfreq(n, d=h) for n, h in zip(self.shape, self.step))

    def fft_mesh(self):
        kx, ky = torch.meshgrid(*self.fft_axes(), indexing="ij")
        return kx.to(self.device), ky.to(self.device)

    def rfft_mesh(self):
        """Half-spectrum wavenumbers: last axis cut to n//2+1 entries, so the
        Nyquist column carries the negative frequency (grids.py:197-201)."""
        k_max = math.floor(self.shape[-1] / 2.0)
        return tuple(k[..., : k_max + 1] for k in self.fft_mesh())

    def __repr__(self):
        return f"Grid(shape={self.shape}, domain={self.domain})"

    def __eq__(self, other):
        return isinstance(other, Grid) and self.shape == other.shape and self.domain == other.domain

    def __hash__(self):
        return hash((self.shape, self.domain))
