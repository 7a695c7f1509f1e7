"""ctypes binding of libtcfd_hip.so (the C ABI declared in include/tcfd.h).

The library is built in-tree by ``build_library()`` (hipcc, gfx950) and loaded
from ``torch-cfd_amd/csrc``.  There is NO fallback: if the shared object is
missing or fails to load, every operator of this package raises.

``import torch`` happens before the dlopen on purpose: PyTorch-ROCm ships its
own ``libamdhip64.so.7``; loading it first makes the dynamic loader bind this
library to the SAME HIP runtime instance (matching SONAME), so device pointers
and stream handles coming from torch are valid inside the kernels' launches.
"""
from __future__ import annotations

import ctypes
import os
import subprocess
import sys
from typing import Optional

import torch  # noqa: F401  (must precede the dlopen, see module docstring)

_HERE = os.path.dirname(os.path.realpath(__file__))   # realpath: the package may be imported through its symlink
CSRC = os.path.join(_HERE, "csrc")
INCLUDE = os.path.join(os.path.dirname(_HERE), "include")
LIB_NAME = "libtcfd_hip.so"
LIB_PATH = os.path.join(CSRC, LIB_NAME)
SOURCES = ("tcfd_ns2d.hip", "tcfd_fno.hip", "tcfd_fno_pw.hip", "tcfd_fno_tiles.hip", "tcfd_loss.hip")

TCFD_C64, TCFD_C128 = 0, 1
ABI_VERSION = 7   # TCFD_ABI_VERSION of include/tcfd.h the SIGNATURES table below was written against

_lib: Optional[ctypes.CDLL] = None


class TcfdError(RuntimeError):
    pass


def _hipcc() -> str:
    for cand in (os.environ.get("HIPCC"), "/opt/rocm/bin/hipcc", "hipcc"):
        if cand and (os.path.isabs(cand) and os.path.exists(cand) or not os.path.isabs(cand)):
            return cand
    return "hipcc"


def build_library(force: bool = False, verbose: bool = False) -> str:
    """Compile csrc/*.hip for gfx950 into csrc/libtcfd_hip.so (cross-compiles
    without a GPU).  Rebuilds only when a source/header is newer than the .so."""
    srcs = [os.path.join(CSRC, s) for s in SOURCES if os.path.exists(os.path.join(CSRC, s))]
    deps = srcs + [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".hpp")]
    deps.append(os.path.join(INCLUDE, "tcfd.h"))
    def fresh():
        return os.path.exists(LIB_PATH) and os.path.getmtime(LIB_PATH) >= max(os.path.getmtime(d) for d in deps)

    if not force and fresh():
        return LIB_PATH
    # several ranks of one node may get here at once (torch.distributed.run): one builds, the others wait for it
    import fcntl

    lock = open(os.path.join(CSRC, ".build.lock"), "w")
    fcntl.flock(lock, fcntl.LOCK_EX)
    try:
        if not force and fresh():
            return LIB_PATH
        return _build_locked(srcs, verbose)
    finally:
        fcntl.flock(lock, fcntl.LOCK_UN)
        lock.close()


# compile jobs: (source, extra flags, object).  tcfd_ns2d.hip is compiled twice -- unit 0 = C ABI + float64 kernels, unit 1 =
# float32 kernels -- and the FNO kernels are three files (transforms + contraction / pointwise block / tiled backward), so that
# the hipcc processes take ~3.5 minutes side by side instead of 7 + 5 one after the other.
JOBS = (("tcfd_ns2d.hip", ("-DTCFD_UNIT=0",), "tcfd_ns2d.o"),
        ("tcfd_ns2d.hip", ("-DTCFD_UNIT=1",), "tcfd_ns2d_f32.o"),
        ("tcfd_fno.hip", (), "tcfd_fno.o"),
        ("tcfd_fno_pw.hip", (), "tcfd_fno_pw.o"),
        ("tcfd_fno_tiles.hip", (), "tcfd_fno_tiles.o"),
        ("tcfd_loss.hip", (), "tcfd_loss.o"))


def _build_locked(srcs, verbose):
    objs = []
    procs = []
    for src, flags, obj in JOBS:
        s, o = os.path.join(CSRC, src), os.path.join(CSRC, obj)
        cmd = [_hipcc(), "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", *flags, "-c", s, "-o", o]
        if verbose:
            print(" ".join(cmd), file=sys.stderr)
        procs.append((cmd, subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT)))
        objs.append(o)
    for cmd, p in procs:
        out, _ = p.communicate()
        if p.returncode != 0:
            raise TcfdError("hipcc failed: %s\n%s" % (" ".join(cmd), out.decode(errors="replace")))
    tmp = LIB_PATH + ".tmp.%d" % os.getpid()
    link = [_hipcc(), "--offload-arch=gfx950", "-shared", "-fPIC", *objs, "-o", tmp, "-Wl,-rpath,/opt/rocm/lib"]
    r = subprocess.run(link, stdout=subprocess.PIPE, stderr=subprocess.STDOUT)
    if r.returncode != 0:
        raise TcfdError("link failed: %s\n%s" % (" ".join(link), r.stdout.decode(errors="replace")))
    os.replace(tmp, LIB_PATH)   # atomic: a process that already mapped the old file keeps a consistent image
    return LIB_PATH


_vp, _i, _l, _d, _sz = ctypes.c_void_p, ctypes.c_int, ctypes.c_long, ctypes.c_double, ctypes.c_size_t
_dp = ctypes.POINTER(ctypes.c_double)

# name -> (restype, argtypes); mirrors include/tcfd.h one to one
SIGNATURES = {
    "tcfd_last_error": (ctypes.c_char_p, []),
    "tcfd_version": (_i, []),
    "tcfd_ns2d_plan_create": (_i, [ctypes.POINTER(_vp), _i, _i, _dp, _dp, _dp, _dp, _dp]),
    "tcfd_ns2d_plan_destroy": (None, [_vp]),
    "tcfd_ns2d_plan_info": (_i, [_vp, ctypes.POINTER(_i), ctypes.POINTER(_i), ctypes.POINTER(_i)]),
    "tcfd_ns2d_plan_variant": (_i, [_vp, ctypes.POINTER(_i), ctypes.POINTER(_i)]),
    "tcfd_ns2d_plan_chunking": (_i, [_vp, _l, ctypes.POINTER(_l), ctypes.POINTER(_sz), ctypes.POINTER(_i)]),
    "tcfd_debug_xl_fft1024": (_i, [_vp, _vp, _vp, _i, _i, _vp]),
    "tcfd_ns2d_workspace_bytes": (_sz, [_vp, _l]),
    "tcfd_ns2d_step": (_i, [_vp, _vp, _vp, _vp, _l, _i, _dp, _dp, _dp, _i, _d, _vp, _sz, _vp]),
    "tcfd_ns2d_step_imex": (_i, [_vp, _vp, _vp, _vp, _l, _i, _dp, _dp, _dp, _dp, _dp, ctypes.POINTER(_i), _i, _d, _vp, _sz,
                                 _vp]),
    "tcfd_ns2d_explicit_terms": (_i, [_vp, _vp, _vp, _l, _vp, _sz, _vp]),
    "tcfd_ns2d_explicit_terms_vjp": (_i, [_vp, _vp, _vp, _vp, _l, _vp, _sz, _vp]),
    "tcfd_ns2d_vjp_combine": (_i, [_vp, _vp, _vp, _l, _l, _i, _vp]),
    "tcfd_ns2d_stage_update": (_i, [_vp, _vp, _vp, _vp, ctypes.POINTER(_d), _vp, _vp, _l, _l, _i, _vp]),
    "tcfd_ns2d_stage_update_vjp": (_i, [_vp, _vp, _vp, ctypes.POINTER(_d), _vp, _vp, _vp, _l, _l, _i, _vp]),
    "tcfd_ns2d_stream_residual": (_i, [_vp, _vp, _vp, _vp, _vp, _l, _vp, _sz, _vp]),
    "tcfd_ns2d_velocity": (_i, [_vp, _vp, _vp, _vp, _vp, _l, _vp]),
    "tcfd_rfft2": (_i, [_vp, _vp, _vp, _l, _vp]),
    "tcfd_irfft2": (_i, [_vp, _vp, _vp, _l, _vp, _sz, _vp]),
    "tcfd_irfft2_subsample": (_i, [_vp, _vp, _vp, _l, _i, _vp, _sz, _vp]),
    "tcfd_irfft2_subsample_max_factor": (_i, [_vp]),
    "tcfd_fno_plan_create": (_i, [ctypes.POINTER(_vp), _i, _i, _i, _i, _i, _i, _i, _i]),
    "tcfd_fno_plan_create_resample": (_i, [ctypes.POINTER(_vp), _i, _i, _i, _i, _i, _i, _i, _i, _i, _i]),
    "tcfd_fno_plan_create_dtype": (_i, [ctypes.POINTER(_vp), _i, _i, _i, _i, _i, _i, _i, _i, _i, _i, _i]),
    "tcfd_fno_plan_destroy": (None, [_vp]),
    "tcfd_fno_workspace_bytes": (_sz, [_vp, _i, _i, _i]),
    "tcfd_fno_spectral_conv": (_i, [_vp, _vp, ctypes.POINTER(_vp), ctypes.POINTER(_vp), _d, _vp, _i, _i,
                                    _i, _i, _d, _d, _i, _vp, _sz, _vp]),
    "tcfd_fno_forward_trunc": (_i, [_vp, _vp, _vp, _i, _i, _d, _vp, _sz, _vp]),
    "tcfd_fno_inverse_trunc": (_i, [_vp, _vp, _vp, _i, _i, _i, _d, _vp, _sz, _vp]),
    "tcfd_fno_inverse_trunc_acc": (_i, [_vp, _vp, _vp, _vp, _i, _i, _i, _d, _vp, _sz, _vp]),
    "tcfd_fno_inverse_trunc_last": (_i, [_vp, _vp, _vp, _vp, _i, _i, _i, _d, _vp, _sz, _vp]),
    "tcfd_fno_forward_trunc_kt": (_i, [_vp, _vp, _vp, _i, _i, _d, _vp, _vp, _sz, _vp]),
    "tcfd_fno_inverse_trunc_kt": (_i, [_vp, _vp, _vp, _vp, _vp, _i, _i, _i, _d, _vp, _vp, _sz, _vp]),
    "tcfd_fno_contract": (_i, [_vp, ctypes.POINTER(_vp), ctypes.POINTER(_vp), _d, _vp, _i, _i, _i, _i, _i,
                               _i, _i, _i, _vp]),
    "tcfd_fno_contract_adjoint": (_i, [_vp, ctypes.POINTER(_vp), _vp, _i, _i, _i, _i, _i, _i, _i, _i, _vp]),
    "tcfd_fno_contract_wgrad": (_i, [_vp, _vp, ctypes.POINTER(_vp), ctypes.POINTER(_vp), _d, _i, _i, _i, _i, _i, _i, _i, _vp]),
    "tcfd_fno_pointwise": (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _l, _i, _i, _i, _i, _i, _l,
                                _l, _vp, _vp]),
    "tcfd_fno_pointwise_pre": (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _l, _i, _i, _i, _i, _i, _l,
                                    _l, _vp, _vp]),
    "tcfd_fno_pointwise_f64": (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _l, _i, _i, _i, _i, _i, _l, _l,
                                    _vp]),
    "tcfd_row_moments_f64": (_i, [_vp, _vp, _i, _l, _vp]),
    "tcfd_sum_rows_slices": (_i, [_l]),
    "tcfd_sum_rows": (_i, [_vp, _vp, _vp, _l, _l, _vp]),
    "tcfd_sum_rows_scatter": (_i, [_vp, _vp, _l, _l, _i, _vp, _vp, _vp]),
    "tcfd_sum_t_into_last": (_i, [_vp, _vp, _l, _i, _i, _vp]),
    "tcfd_row_moments": (_i, [_vp, _vp, _i, _l, _vp]),
    "tcfd_fno_pointwise_bwd": (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, ctypes.POINTER(_i),
                                    _i, _i, _i, _i, _l, _i, _i, _i, _i, _i, _i, _vp]),
    "tcfd_fno_pointwise_bwd_out": (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, ctypes.POINTER(_i),
                                        _i, _i, _i, _i, _l, _i, _i, _i, _i, _i, _i, _vp]),
    "tcfd_fno_lift_spectrum": (_i, [_vp, _vp, _vp, _vp, _vp, _i, _i, _i, _l, _vp]),
    "tcfd_fno_sample_outer_sums": (_i, [_vp, _vp, _vp, _vp, _i, _i, _i, _l, _i, _vp]),
    "tcfd_fno_plan_supports": (_i, [_vp, _i]),
    "tcfd_fno_profile_begin": (_i, [_i]),
    "tcfd_fno_profile_end": (_i, [_i, ctypes.POINTER(_i), ctypes.POINTER(_i), ctypes.POINTER(ctypes.c_float)]),
    "tcfd_fno_pointwise_bwd_saved": (_i, [_i, _i, _i, _l, _i, _i]),
    "tcfd_fno_pointwise_bwd_pe": (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, ctypes.POINTER(_i), _i, _i, _i, _l, _i, _vp]),
    "tcfd_ns2d_profile_begin": (_i, [_vp, _i]),
    "tcfd_ns2d_profile_end": (_i, [_vp, _i, ctypes.POINTER(_i), ctypes.POINTER(_i), ctypes.POINTER(ctypes.c_float)]),
    "tcfd_weighted_sqnorm": (_i, [_vp, _vp, _vp, _l, _l, _i, _i, _vp]),
    "tcfd_fno_reduce_frames": (_i, [_vp, _vp, _vp, _vp, _vp, _i, _i, _i, _l, _i, _vp]),
    "tcfd_fno_inverse_trunc_residual": (_i, [_vp, _vp, _vp, _vp, _i, _i, _i, _i, _d, _vp, _sz, _vp]),
    "tcfd_fno_lift_fold": (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _d, _vp, _vp, _vp, _vp, _i, _i, _i, _l, _vp]),
    "tcfd_loss_plan_create": (_i, [ctypes.POINTER(_vp), _i, _i]),
    "tcfd_loss_plan_destroy": (None, [_vp]),
    "tcfd_loss_workspace_bytes": (_sz, [_vp, _l, _i, _i]),
    "tcfd_sobolev_loss_supported": (_i, [_vp, _i, _i]),
    "tcfd_sobolev_loss": (_i, [_vp, _vp, _vp, _vp, _l, _i, _i, _i, _i, _i, _i, _vp, _vp, _vp, _sz, _vp]),
    "tcfd_sobolev_loss_backward": (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _l, _i, _i, _i, _i, _i, _i, _vp, _vp, _sz, _vp]),
    "tcfd_hbm_probe": (_i, [_vp, _vp, ctypes.c_size_t, _i, _i, ctypes.POINTER(ctypes.c_float), _vp]),
    "tcfd_copy_rows_to_host": (_i, [_vp, _sz, _vp, _sz, _sz, _sz, _vp]),
    "tcfd_host_register": (_i, [_vp, _sz]),
    "tcfd_host_unregister": (_i, [_vp]),
}


def load() -> ctypes.CDLL:
    """dlopen the library (once) and declare every prototype of include/tcfd.h."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise TcfdError(
            f"{LIB_PATH} not found: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            "(hipcc --offload-arch=gfx950). torch-cfd_amd has no CPU / eager fallback.")
    lib = ctypes.CDLL(LIB_PATH)
    lib.tcfd_version.restype, lib.tcfd_version.argtypes = _i, []
    have = lib.tcfd_version()
    if have != ABI_VERSION:
        # a stale prebuilt library (newer mtime than the sources, older argument lists): calling it would pass doubles where
        # it reads floats.  Never guess -- say so; build_library(force=True) replaces it.
        raise TcfdError(f"{LIB_PATH} has ABI revision {have}, this package expects {ABI_VERSION} (include/tcfd.h TCFD_ABI_VERSION): "
                        "rebuild it with `python -c 'import __graft_entry__ as g; g.build()'` after deleting the stale file, or "
                        "torch_cfd_amd._lib.build_library(force=True)")
    for name, (res, args) in SIGNATURES.items():
        if not hasattr(lib, name):
            continue  # optional symbol groups are checked by tests/test_abi.py against the header
        fn = getattr(lib, name)
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib


def check(rc: int, what: str = "") -> None:
    if rc != 0:
        msg = load().tcfd_last_error()
        raise TcfdError(f"{what} failed (status {rc}): {msg.decode() if msg else ''}")


def darray(values) -> ctypes.Array:
    vals = [float(v) for v in values]
    return (ctypes.c_double * len(vals))(*vals)


def dptr_of_tensor(t: "torch.Tensor"):
    """Host double pointer of a contiguous float64 CPU tensor."""
    assert t.dtype == torch.float64 and t.device.type == "cpu" and t.is_contiguous()
    return ctypes.cast(t.data_ptr(), _dp)


def current_stream_ptr(device) -> int:
    return torch.cuda.current_stream(device).cuda_stream
