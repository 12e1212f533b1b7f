"""Drop-in for the reference module of the same name (ShanChen2D/AccelerateGPU2D.py): the explicit-forcing pipeline of the legacy DENSE
kernel file (:1336-2487, calHalfWallBounceBack :2698, and the two macro kernels :54 / :80 its driver chains in front) as pre-built HIP
kernels on the file's own dense direction-major arrays f[9][ny * nx] with boolean masks (include/lbmpm_kernels.h, lbmpm_de_*;
csrc/dense_ef.h), callable as kernel[grid, block](...).  The file's two quirks are replicated (the equilibrium of :2354 with its
direction-7 typo, v_y of :92 not divided by the density).  Known answers: tests/golden/dense_kernels.npz, tests/test_dense_dropin_gpu.py."""
import os as _os
import sys as _sys

_sys.path.insert(0, _os.path.dirname(_os.path.abspath(__file__)))
from _runtime import export as _export  # noqa: E402

_export("de", globals())
