"""`numba.cuda` facade over liblbmpm_hip.so (device arrays + availability probes).  `cuda.jit` is
deliberately absent: the kernels are the pre-built HIP ones exported by the sibling modules
(AcceleratedRKGPU2D, OptimizedD2Q9GPU, ExplicitD2Q9GPU, AccelerateTransport2DRK)."""
import os as _os
import sys as _sys

_sys.path.insert(0, _os.path.dirname(_os.path.dirname(_os.path.abspath(__file__))))
from _runtime import DeviceNDArray, to_device, device_array_like, device_array, _L  # noqa: E402,F401


def is_available():
    return _L().lbmpm_device_count() > 0


def detect():
    n = _L().lbmpm_device_count()
    print("Found %d HIP device(s) (gfx950 expected)" % max(n, 0))
    return n > 0


class _Gpus(list):
    def __repr__(self):
        return "<HIP devices: %d>" % len(self)


gpus = _Gpus(range(max(_L().lbmpm_device_count(), 0)))


def synchronize():
    from openlbmpm_amd import _lib
    _lib.check(_L().lbmpm_device_synchronize(), "synchronize")


def jit(*a, **k):
    raise NotImplementedError("this is not a JIT: import the pre-built HIP kernels from AcceleratedRKGPU2D / "
                              "OptimizedD2Q9GPU / ExplicitD2Q9GPU / AccelerateTransport2DRK (openlbmpm_amd/dropin)")
