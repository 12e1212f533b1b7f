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


def rel_err(a, g):
    """max |a-g| / max |g|  (field-relative, the north star's 'relative' tolerance)."""
    a = np.asarray(a, dtype=np.float64); g = np.asarray(g, dtype=np.float64)
    if a.shape != g.shape:
        raise AssertionError("shape %s vs %s" % (a.shape, g.shape))
    if not np.all(np.isfinite(a) == np.isfinite(g)):
        return np.inf
    m = np.isfinite(g)
    scale = max(float(np.max(np.abs(g[m]))) if m.any() else 0.0, 1e-300)
    return float(np.max(np.abs(a[m] - g[m]))) / scale if m.any() else 0.0
