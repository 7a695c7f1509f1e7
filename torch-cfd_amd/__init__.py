"""torch-cfd_amd: MI355X-native (gfx950) implementation of torch-cfd's spectral hot path.

Operator API (same names/signatures as the reference) over hand-written HIP
kernels reached through the C ABI in ``include/tcfd.h``.  No CPU fallback.
The directory name carries a hyphen (the repository layout asks for it); the
importable name ``torch_cfd_amd`` is a symbolic link to this very directory, so
there is ONE package with one module identity.
"""
from . import _lib  # noqa: F401
from .grids import Grid  # noqa: F401
from .equations import (  # noqa: F401
    IMEXStepper,
    run_stage_schedule,
    ImplicitExplicitODE,
    NavierStokes2DSpectral,
    RK4CrankNicolsonStepper,
    fft_plan,
    stable_time_step,
)
from .forcings import FieldArray, ForcingFn, KolmogorovForcing, SimpleSolenoidalForcing, SinCosForcing  # noqa: F401
from .solvers import get_trajectory_imex  # noqa: F401
from .spectral import (  # noqa: F401
    brick_wall_filter_2d,
    fft_mesh_2d,
    spectral_curl_2d,
    spectral_div_2d,
    spectral_grad_2d,
    spectral_laplacian_2d,
    spectral_rot_2d,
    vorticity_to_velocity,
)

__version__ = "0.1.0"
