"""`numba` as seen by a reference driver when openlbmpm_amd/dropin is first on PYTHONPATH: only the
names the reference imports (jit/autojit decorators as no-ops on host helpers, dtype aliases, cuda)."""
import numpy as _np

from . import cuda  # noqa: F401

float64, float32, int64, int32, boolean = _np.float64, _np.float32, _np.int64, _np.int32, _np.bool_


def _identity(*dargs, **dkwargs):
    if len(dargs) == 1 and callable(dargs[0]) and not dkwargs:
        return dargs[0]
    return lambda fn: fn


jit = autojit = njit = _identity


def vectorize(*dargs, **dkwargs):
    if len(dargs) == 1 and callable(dargs[0]) and not dkwargs:
        return _np.vectorize(dargs[0])
    return lambda fn: _np.vectorize(fn)
