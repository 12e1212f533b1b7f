"""Container-only harness that makes /root/reference importable and runnable.

Golden vectors under tests/golden/*.npz are produced by importing the real reference
modules through this harness (numba stand-in in ./numba, stubs below) and running the
reference's own drivers / kernels.  Only the *outputs* (arrays) are committed; the
reference sources never leave /root/reference.  See SURVEY.md Appendix C.

Nothing under tests/golden/gen is imported by the test-suite or the product.
"""
import builtins
import os
import sys
import types
import tempfile

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
REF = os.environ.get("LBMPM_REFERENCE", "/root/reference")

H5_CAPTURE = {}     # "<file>:/<group>/<name>" -> ndarray   (filled by the tables stub)


def _install_tables_stub():
    mod = types.ModuleType("tables")

    class _Node:
        pass

    class _File:
        def __init__(self, path, mode='r'):
            self.path = os.path.basename(path)
            self.root = _Node()

        def create_group(self, where, name, title=''):
            return None

        def create_array(self, where, name, obj=None, title=''):
            w = where if isinstance(where, str) else "/"
            H5_CAPTURE["%s:%s/%s" % (self.path, w.rstrip('/'), name)] = np.array(obj, copy=True)

        def close(self):
            return None

    mod.open_file = lambda path, mode='r', **kw: _File(path, mode)
    sys.modules["tables"] = mod


def _install_scipy_aliases():
    import scipy as sp
    for name in ("empty", "arange", "sqrt", "zeros", "ones", "array", "power", "sum",
                 "float64", "int64", "exp", "log", "pi", "cos", "sin"):
        if not hasattr(sp, name):
            setattr(sp, name, getattr(np, name))


def _install_matplotlib_noops():
    import matplotlib
    matplotlib.use("Agg")
    import matplotlib.pyplot as plt
    for name in ("imshow", "colorbar", "savefig", "close", "figure", "subplot", "plot",
                 "show", "title", "contour", "contourf", "quiver"):
        setattr(plt, name, lambda *a, **k: None)


def setup(extra_paths=("RKCG2D", "ShanChen2D")):
    """Idempotent: prepare sys.path / sys.modules so reference modules import."""
    if HERE not in sys.path:
        sys.path.insert(0, HERE)            # ./numba stand-in wins
    for p in extra_paths:
        full = os.path.join(REF, p)
        if full not in sys.path:
            sys.path.append(full)
    import numba  # noqa: F401  (the stand-in)
    from numba import cuda
    _install_tables_stub()
    _install_scipy_aliases()
    _install_matplotlib_noops()
    builtins.input = lambda *a, **k: ''
    builtins.cuda = cuda                    # RKGPU2DBoundary.py has no imports at all
    import getpass
    getpass.getuser = lambda: "oracle"
    # quiet the per-launch prints of the reference drivers
    if os.environ.get("LBMPM_REF_VERBOSE", "0") != "1":
        builtins.print = _quiet_print
    # RKD2Q9.py:21 imports a module that is not shipped; the only geometry helper in the
    # tree is ShanChen2D/SimpleGeometry.py (same function name).
    import SimpleGeometry
    sys.modules.setdefault("SimpleGeometryRK", SimpleGeometry)
    import math
    _acos = math.acos

    def _safe_acos(x):          # CUDA returns NaN outside [-1,1]; Python raises
        if x > 1.0 or x < -1.0:
            return float(np.arccos(np.float64(x)))
        return _acos(x)
    math.acos = _safe_acos
    return cuda


_real_print = builtins.print


def _quiet_print(*a, **k):
    return None


def say(*a, **k):
    _real_print(*a, **k)
    sys.stdout.flush()


def write_ini_dir(files):
    """files: {filename: text}; returns a fresh directory path containing them."""
    d = tempfile.mkdtemp(prefix="lbmpm_ini_")
    for name, text in files.items():
        with open(os.path.join(d, name), "w") as fh:
            fh.write(text)
    return d
