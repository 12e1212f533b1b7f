"""ctypes front-end of oracle/rk3d_csf_oracle.c (D3Q19 colour gradient with CSF tension: the reference's 2-D CSF loop carried to 3-D,
pinned by reduction to it -- tests/test_oracle_rk3d_csf.py).  TEST INFRASTRUCTURE ONLY -- see oracle/__init__.py."""
import ctypes as C

import numpy as np

from . import lib

F64P = C.POINTER(C.c_double)
U8P = C.POINTER(C.c_uint8)

_F19 = ("fR", "fB", "gR", "gB", "fT")
_F1 = ("rhoR", "rhoB", "vx", "vy", "vz", "phi", "Gx", "Gy", "Gz", "Fx", "Fy", "Fz", "K", "nsx", "nsy", "nsz")


class _Sim(C.Structure):
    _fields_ = [("nx", C.c_int64), ("ny", C.c_int64), ("nz", C.c_int64), ("dom", U8P)] + \
               [(n, C.c_double) for n in ("sigma", "cosT", "sinT", "beta", "delta", "tauR", "tauB", "vzIn", "pInB", "pInR", "pOutTotal")] + \
               [(n, C.c_int) for n in ("tauType", "mrt", "inletType", "outletType", "wetting")] + \
               [(n, F64P) for n in _F19 + _F1] + \
               [("kind", U8P), ("W", C.c_int64), ("rates", C.c_double * 6), ("crisp", C.c_double)]


# the 2-D ini's parameters (RKtwophasesetup2D.ini) under the 3-D ini's key names for the flow axis
DEFAULT_PARAMS = dict(sigma=0.1, theta=60.0, wetting=2, beta=0.7, delta=0.98, tauR=1.0, tauB=1.0, tautype=2, relax="MRT",
                      inlet="Neumann", outlet="Dirichlet", velocityZR=-1.0e-4, velocityZB=0.0, densityBH=5e-8, densityRH=1.00536,
                      densityBL=1.0, densityRL=5e-8, rates=(1.19, 1.4, 1.2, 1.4, 1.2, 0.0), crisp=0.0)


class RK3DCSFOracle:
    def __init__(self, dom, rhoR0, rhoB0, params=None, velocity=None):
        L = lib()
        p = dict(DEFAULT_PARAMS); p.update(params or {})
        self.p = p
        self.dom = np.ascontiguousarray(dom, dtype=np.uint8)
        nz, ny, nx = self.dom.shape
        N = nz * ny * nx
        self.shape = (nz, ny, nx)
        for name in _F19:
            setattr(self, "_" + name, np.zeros((N, 19)))
        for name in _F1:
            setattr(self, "_" + name, np.zeros(N))
        self._kind = np.zeros(N, dtype=np.uint8)
        s = _Sim()
        s.nx, s.ny, s.nz = nx, ny, nz
        s.dom = self.dom.ctypes.data_as(U8P)
        th = p["theta"] / 180. * np.pi
        s.sigma, s.cosT, s.sinT = p["sigma"], float(np.cos(th)), float(np.sin(th))
        s.beta, s.delta, s.tauR, s.tauB = p["beta"], p["delta"], p["tauR"], p["tauB"]
        s.vzIn = p["velocityZB"] + p["velocityZR"]
        s.pInB, s.pInR = p["densityBH"], p["densityRH"]
        s.pOutTotal = p["densityBL"] + p["densityRL"]
        s.tauType, s.mrt = int(p["tautype"]), int(p["relax"] == "MRT")
        s.inletType, s.outletType = int(p["inlet"] == "Dirichlet"), int(p["outlet"] == "Convective")
        s.wetting = int(p["wetting"])
        for i, r in enumerate(p["rates"]):
            s.rates[i] = float(r)
        s.crisp = float(p["crisp"])
        for name in _F19 + _F1:
            setattr(s, name, getattr(self, "_" + name).ctypes.data_as(F64P))
        s.kind = self._kind.ctypes.data_as(U8P)
        self._s, self._L = s, L
        L.rk3dcsf_setup(C.byref(s))
        a = np.ascontiguousarray(rhoR0, dtype=np.float64); b = np.ascontiguousarray(rhoB0, dtype=np.float64)
        v = [None if velocity is None or c is None else np.ascontiguousarray(c, dtype=np.float64) for c in (velocity or (None, None, None))]
        vp = [C.cast(None, F64P) if c is None else c.ctypes.data_as(F64P) for c in v]
        L.rk3dcsf_init(C.byref(s), a.ctypes.data_as(F64P), b.ctypes.data_as(F64P), *vp)

    @property
    def W(self):
        return int(self._s.W)

    def set_populations(self, fR, fB, force=None):
        """the loop's arrays at its top from streamed populations [nz][ny][nx][19] per colour (+ the force of the step before)"""
        a = np.ascontiguousarray(fR, dtype=np.float64); b = np.ascontiguousarray(fB, dtype=np.float64)
        self._L.rk3dcsf_set_populations(C.byref(self._s), a.ctypes.data_as(F64P), b.ctypes.data_as(F64P))
        if force is not None:
            for n, c in zip(("Fx", "Fy", "Fz"), force):
                np.ctypeslib.as_array(getattr(self._s, n), shape=(int(np.prod(self.shape)),))[:] = np.asarray(c, dtype=np.float64).reshape(-1)
        return self

    def run(self, n):
        self._L.rk3dcsf_run(C.byref(self._s), C.c_int64(int(n)))
        return self

    def step_a(self):
        self._L.rk3dcsf_step_a(C.byref(self._s))
        return self

    def field(self, name):
        if name == "kind":
            return self._kind.reshape(self.shape).copy()
        ptr = getattr(self._s, name)       # populations swap buffers inside the C struct
        n = int(np.prod(self.shape)) * (19 if name in _F19 else 1)
        a = np.ctypeslib.as_array(ptr, shape=(n,)).copy()
        return a.reshape(self.shape + ((19,) if name in _F19 else ()))
