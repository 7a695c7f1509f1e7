"""Import shim: ``import torch_cfd_amd`` -> the package in ``../torch-cfd_amd/``
(a hyphenated directory name is not importable by itself)."""
import os as _os

_real = _os.path.join(_os.path.dirname(_os.path.dirname(_os.path.abspath(__file__))), "torch-cfd_amd")
__path__[:] = [_real]
with open(_os.path.join(_real, "__init__.py")) as _f:
    exec(compile(_f.read(), _os.path.join(_real, "__init__.py"), "exec"))
del _f
