"""Shared helpers for the parity tests (oracle = checker; never the thing under test on
the product side)."""
import glob
import os

import numpy as np

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def golden_files(prefix):
    return sorted(glob.glob(os.path.join(GOLDEN, prefix + "*.npz")))


def load_params(d):
    return {k[4:]: d[k].item() for k in d.files if k.startswith("par_")}


def rel_err(a, g, scale=None):
    """max |a-g| / max |g|  (field-relative, the north star's 'relative' tolerance).  `scale` replaces max |g|:
    the components of a vector field are measured against the magnitude of the vector, not each against itself
    (a component that is zero but for round-off has no scale of its own)."""
    a = np.asarray(a, dtype=np.float64); g = np.asarray(g, dtype=np.float64)
    if a.shape != g.shape:
        raise AssertionError("shape %s vs %s" % (a.shape, g.shape))
    if not np.all(np.isfinite(a) == np.isfinite(g)):
        return np.inf
    m = np.isfinite(g)
    scale = float(scale) if scale is not None else max(float(np.max(np.abs(g[m]))) if m.any() else 0.0, 1e-300)
    return float(np.max(np.abs(a[m] - g[m]))) / scale if m.any() else 0.0
