"""Periodic box descriptor handed to the spectral operator.

The spectral path needs three things from a grid: the sample counts, the box
(for the wavenumbers ``k = index / length``) and sample positions (for forcings
and initial conditions).  ``Grid`` keeps the constructor keywords and the
attribute / method names the reference's callers use (``Grid(shape=, domain=,
step=, device=)``, ``.shape .domain .step .ndim .mesh() .fft_axes() .fft_mesh()
.rfft_mesh()``, torch_cfd/grids.py:37-201) so that scripts written against the
reference construct the operator unchanged; the staggered-grid data model behind
the reference class (offsets per variable, GridArray / GridVariable) belongs to
its finite-volume solver and is not part of this path.
"""
from __future__ import annotations

import torch


def _as_box(counts, step, domain):
    """((lo, hi), ...) per axis from either a cell size or an extent."""
    d = len(counts)
    if step is not None and domain is not None:
        raise TypeError("Grid: give the cell size (`step`) or the box (`domain`), not both")
    if domain is None:
        h = torch.as_tensor(1.0 if step is None else step, dtype=torch.float64).flatten()
        if h.numel() not in (1, d):
            raise ValueError(f"Grid: `step` has {h.numel()} entries for a {d}-D grid")
        h = h.expand(d)
        return tuple((0.0, float(h[a]) * counts[a]) for a in range(d))
    box = torch.as_tensor(domain, dtype=torch.float64)
    if box.ndim == 0:  # a single length: the box [0, L]^d
        return tuple((0.0, float(box)) for _ in range(d))
    if tuple(box.shape) != (d, 2):
        raise ValueError(f"Grid: `domain` must be one length or {d} (lo, hi) pairs, got shape {tuple(box.shape)}")
    return tuple((float(lo), float(hi)) for lo, hi in box.tolist())


class Grid:
    """``shape`` samples on the periodic box ``domain``; sample i of axis a sits at
    ``lo_a + (i + offset_a) * step_a`` (offset 1/2 = cell centres, 0 = cell corners)."""

    def __init__(self, shape, step=None, domain=None, device="cpu"):
        self.shape = tuple(int(n) for n in shape)
        if any(n <= 0 for n in self.shape):
            raise ValueError(f"Grid: sample counts must be positive, got {self.shape}")
        self.domain = _as_box(self.shape, step, domain)
        # the same expression the wavenumber spacing is built from: (hi - lo) / n
        self.step = tuple((hi - lo) / n for (lo, hi), n in zip(self.domain, self.shape))
        self.device = device

    # -- geometry
    @property
    def ndim(self):
        return len(self.shape)

    @property
    def cell_center(self):
        """Offset (in cells) of the cell centres: 1/2 along every axis."""
        return (0.5,) * self.ndim

    @property
    def cell_faces(self):
        """Per axis a, the offset of the face a cell shares with its upper neighbour along a: 1 there, 1/2 elsewhere
        (the staggered positions the reference's forcings default to, torch_cfd/grids.py:117-121)."""
        return tuple(tuple(1.0 if b == a else 0.5 for b in range(self.ndim)) for a in range(self.ndim))

    def axes(self, offset=None):
        """1-D sample positions per axis, shifted by ``offset`` cells (default: cell centres)."""
        offset = self.cell_center if offset is None else tuple(offset)
        if len(offset) != self.ndim:
            raise ValueError(f"Grid.axes: offset {offset} for a {self.ndim}-D grid")
        return tuple(lo + (torch.arange(n) + o) * h for (lo, _), n, o, h in zip(self.domain, self.shape, offset, self.step))

    def mesh(self, offset=None):
        """Coordinate arrays (``indexing='ij'``) of the samples shifted by ``offset`` cells (default 1/2)."""
        return tuple(c.to(self.device) for c in torch.meshgrid(*self.axes(offset), indexing="ij"))

    # -- wavenumbers (ordinal: cycles per unit length; the kernels multiply by 2 pi)
    def fft_axes(self):
        return tuple(torch.fft.fftfreq(count, d=width) for count, width in zip(self.shape, self.step))

    def fft_mesh(self):
        return tuple(k.to(self.device) for k in torch.meshgrid(*self.fft_axes(), indexing="ij"))

    def rfft_mesh(self):
        """Wavenumbers of the half spectrum (last axis: entries 0 .. n//2).  The last kept column is the
        Nyquist one and carries the NEGATIVE frequency -n/2/L, because it is cut out of the full
        ``fftfreq`` ordering -- the c2r semantics of the kernels rely on exactly this table."""
        keep = self.shape[-1] // 2 + 1
        return tuple(k[..., :keep] for k in self.fft_mesh())

    # -- value semantics
    def _key(self):
        return (self.shape, self.domain)

    def __eq__(self, other) -> bool:
        return isinstance(other, Grid) and self._key() == other._key()

    def __hash__(self) -> int:
        return hash(self._key())

    def __repr__(self) -> str:
        return f"Grid(shape={self.shape}, domain={self.domain})"
