"""`numba.cuda` stand-in: sequential SIMT emulation (one Python call per thread).

Semantics that matter for golden capture
  * ``kernel[grid, block](*args)`` runs blocks by -> bx, threads ty -> tx, in order.
    Racy reference kernels therefore get deterministic per-thread semantics
    (SURVEY.md Appendix C.6).
  * ``shared.array`` / ``local.array`` return a fresh zero array per thread call.
  * ``to_device`` returns a private copy whose ``copy_to_host()`` copies back; every
    device array is recorded in ``REGISTRY`` so a capture script can dump the final
    device state of a driver regardless of its hard-coded save cadence.
"""
import types as _types
import numpy as _np

REGISTRY = []          # every array handed out by to_device/device_array*
LAUNCH_LOG = []        # (kernel name, grid, block) per launch, optional
LOG_LAUNCHES = False
PRE_LAUNCH_HOOK = None  # callable(name, args) -> None, optional
POST_LAUNCH_HOOK = None  # callable(name, args) -> None, optional


class _Dim3:
    __slots__ = ("x", "y", "z")

    def __init__(self, x=0, y=0, z=0):
        self.x, self.y, self.z = x, y, z


threadIdx = _Dim3()
blockIdx = _Dim3()
blockDim = _Dim3(1, 1, 1)
gridDim = _Dim3(1, 1, 1)


class DeviceArray(_np.ndarray):
    def copy_to_host(self, ary=None, stream=0):
        if ary is not None:
            ary[...] = self
            return ary
        return _np.array(self, copy=True).view(_np.ndarray)

    def copy_to_device(self, src, stream=0):
        self[...] = src


def _as_device(a):
    d = _np.array(a, copy=True).view(DeviceArray)
    REGISTRY.append(d)
    return d


def to_device(ary, stream=0, copy=True, to=None):
    return _as_device(_np.asarray(ary))


def device_array_like(ary, stream=0):
    return _as_device(_np.zeros_like(_np.asarray(ary)))


def device_array(shape, dtype=_np.float64, strides=None, order='C', stream=0):
    return _as_device(_np.zeros(shape, dtype=dtype))


class _ArrayFactory:
    @staticmethod
    def array(shape, dtype):
        return _np.zeros(shape, dtype=dtype)


shared = _ArrayFactory()
local = _ArrayFactory()
const = _types.SimpleNamespace(array_like=lambda a: _np.array(a))


def syncthreads():
    return None


def grid(ndim):
    if ndim == 1:
        return blockIdx.x * blockDim.x + threadIdx.x
    return (blockIdx.x * blockDim.x + threadIdx.x,
            blockIdx.y * blockDim.y + threadIdx.y)


def _dims(v):
    if isinstance(v, (int, _np.integer)):
        return (int(v), 1, 1)
    v = tuple(int(i) for i in v)
    return v + (1,) * (3 - len(v))


class _Kernel:
    def __init__(self, fn):
        self.py_func = fn
        self.__name__ = fn.__name__
        self.__doc__ = fn.__doc__

    def __getitem__(self, cfg):
        g, b = _dims(cfg[0]), _dims(cfg[1])
        fn = self.py_func
        name = self.__name__

        def launch(*args):
            if LOG_LAUNCHES:
                LAUNCH_LOG.append((name, g, b))
            if PRE_LAUNCH_HOOK is not None:
                PRE_LAUNCH_HOOK(name, args)
            blockDim.x, blockDim.y, blockDim.z = b
            gridDim.x, gridDim.y, gridDim.z = g
            for by in range(g[1]):
                blockIdx.y = by
                for bx in range(g[0]):
                    blockIdx.x = bx
                    for ty in range(b[1]):
                        threadIdx.y = ty
                        for tx in range(b[0]):
                            threadIdx.x = tx
                            fn(*args)
            if POST_LAUNCH_HOOK is not None:
                POST_LAUNCH_HOOK(name, args)
        return launch

    def __call__(self, *args):  # device-function style direct call
        return self.py_func(*args)


def jit(*dargs, **dkwargs):
    device = bool(dkwargs.get('device', False))

    def wrap(fn):
        return fn if device else _Kernel(fn)

    if len(dargs) == 1 and callable(dargs[0]) and not isinstance(dargs[0], str):
        return wrap(dargs[0])
    return wrap


def reduce(fn):
    def run(arr, init=0):
        acc = init
        for v in _np.asarray(arr).ravel():
            acc = fn(acc, v)
        return acc
    return run


def is_available():
    return True


def detect():
    return True


gpus = ["emulated-gpu-0"]


def select_device(i):
    return None


def synchronize():
    return None
