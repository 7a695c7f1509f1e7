"""CPU-only: the C-ABI library builds/loads and exports every symbol include/tcfd.h
declares; the ctypes table mirrors the header.  No compute calls (no GPU here)."""
import os
import re

import pytest

from conftest import ROOT


def header_symbols():
    text = open(os.path.join(ROOT, "include", "tcfd.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(tcfd_[a-z0-9_]+)\s*\(", text)))


def test_header_declares_symbols():
    syms = header_symbols()
    assert "tcfd_ns2d_step" in syms and "tcfd_ns2d_plan_create" in syms and len(syms) >= 12


def test_library_exports_every_header_symbol():
    import torch_cfd_amd as tc

    lib = tc._lib.load()
    for name in header_symbols():
        assert hasattr(lib, name), f"{name} declared in include/tcfd.h but not exported"
    assert lib.tcfd_version() == tc._lib.ABI_VERSION


def test_ctypes_table_matches_header():
    import torch_cfd_amd as tc

    assert sorted(tc._lib.SIGNATURES) == header_symbols()


def test_only_one_hip_runtime_is_mapped():
    """The library must bind to the HIP runtime torch already loaded (same SONAME)."""
    import torch_cfd_amd as tc

    tc._lib.load()
    maps = open("/proc/self/maps").read()
    hips = set(re.findall(r"(/\S*libamdhip64\S*)", maps))
    assert len(hips) == 1, hips


def test_plan_create_rejects_bad_sizes_without_touching_the_gpu():
    import ctypes

    import torch_cfd_amd as tc

    lib = tc._lib.load()
    h = ctypes.c_void_p()
    one = tc._lib.darray([0.0] * 8)
    rc = lib.tcfd_ns2d_plan_create(ctypes.byref(h), 12, 1, one, one, one, one, None)
    assert rc == -1 and b"power of two" in lib.tcfd_last_error()
    rc = lib.tcfd_ns2d_plan_create(ctypes.byref(h), 16, 7, one, one, one, one, None)
    assert rc == -1


def test_header_is_plain_c():
    """include/tcfd.h is the drop-in boundary: it must compile as C99 on its own (no C++ or torch types)."""
    import shutil
    import subprocess

    gcc = shutil.which("gcc")
    if gcc is None:
        pytest.skip("no gcc")
    hdr = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "include", "tcfd.h")
    r = subprocess.run([gcc, "-fsyntax-only", "-x", "c", "-std=c99", "-Wall", "-Wextra", "-Werror", hdr],
                       stdout=subprocess.PIPE, stderr=subprocess.STDOUT)
    assert r.returncode == 0, r.stdout.decode()
