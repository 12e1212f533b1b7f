"""Drop-in for the reference module of the same name: the kernels a working reference driver loop
launches, as pre-built HIP kernels (include/lbmpm_kernels.h), callable as kernel[grid, block](...)."""
import os as _os
import sys as _sys

_sys.path.insert(0, _os.path.dirname(_os.path.abspath(__file__)))
from _runtime import export as _export  # noqa: E402

_export("tr", globals())
