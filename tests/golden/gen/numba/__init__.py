"""Container-only stand-in for the `numba` package (NOT a reimplementation of numba).

Purpose: let the *unmodified* reference modules under /root/reference be imported and
their ``@cuda.jit`` kernels executed, thread by thread, in pure Python so that golden
vectors can be captured from the real reference code.  This is
test tooling that only ever runs in the authoring container; nothing here ships on the
product path and nothing here is derived from numba's sources.
"""
import numpy as _np
from . import cuda  # noqa: F401

float64 = _np.float64
float32 = _np.float32
int64 = _np.int64
int32 = _np.int32
boolean = _np.bool_


def _identity_decorator(*dargs, **dkwargs):
    # used both as ``@jit`` and ``@jit(...)``
    if len(dargs) == 1 and callable(dargs[0]) and not dkwargs:
        return dargs[0]

    def wrap(fn):
        return fn
    return wrap


jit = _identity_decorator
autojit = _identity_decorator
njit = _identity_decorator


def vectorize(*dargs, **dkwargs):
    def wrap(fn):
        return _np.vectorize(fn)
    if len(dargs) == 1 and callable(dargs[0]) and not dkwargs:
        return _np.vectorize(dargs[0])
    return wrap
