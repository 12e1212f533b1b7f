"""ctypes front-end of oracle/rk3d_oracle.c (D3Q19 colour gradient; pinned by reduction to the reference's D2Q9 perturbation loop, tests/test_rk3d_reduction.py).
TEST INFRASTRUCTURE ONLY -- see oracle/__init__.py."""
import ctypes as C

import numpy as np

from . import lib

F64P = C.POINTER(C.c_double)
U8P = C.POINTER(C.c_uint8)


class _Sim(C.Structure):
    _fields_ = [("nx", C.c_int64), ("ny", C.c_int64), ("nz", C.c_int64), ("dom", U8P)] + \
               [(n, C.c_double) for n in ("akR", "akB", "beta", "tauR", "tauB", "solidPhi", "vzR", "vzB",
                                          "rhoOutR", "rhoOutB")] + \
               [("mrt", C.c_int)] + \
               [(n, F64P) for n in ("fR", "fB", "gR", "gB", "rhoR", "rhoB", "phi", "vx", "vy", "vz", "Gx", "Gy", "Gz")] + \
               [("rcA", C.c_double), ("rcD", C.c_double), ("inletP", C.c_int), ("rhoInR", C.c_double), ("rhoInB", C.c_double), ("conv", C.c_int)]


DEFAULT_PARAMS = dict(AkR=7.0e-3, AkB=7.0e-3, beta=1.0, tauR=1.0, tauB=1.0, SolidRhoR=0.7, SolidRhoB=0.0,
                      velocityZR=0.0, velocityZB=-1.0e-4, densityRL=1.0e-8, densityBL=1.0, relax="SRT",
                      inlet="Neumann", densityRH=1.0e-8, densityBH=1.0, outlet="Dirichlet")


class RK3DOracle:
    def __init__(self, dom, rhoR0, rhoB0, params=None):
        L = lib()
        p = dict(DEFAULT_PARAMS); p.update(params or {})
        self.dom = np.ascontiguousarray(dom, dtype=np.uint8)
        nz, ny, nx = self.dom.shape
        N = nz * ny * nx
        self.shape = (nz, ny, nx)
        for name in ("fR", "fB", "gR", "gB"):
            setattr(self, "_" + name, np.zeros((N, 19)))
        for name in ("rhoR", "rhoB", "phi", "vx", "vy", "vz", "Gx", "Gy", "Gz"):
            setattr(self, "_" + name, np.zeros(N))
        s = _Sim()
        s.nx, s.ny, s.nz = nx, ny, nz
        s.dom = self.dom.ctypes.data_as(U8P)
        s.akR, s.akB, s.beta, s.tauR, s.tauB = p["AkR"], p["AkB"], p["beta"], p["tauR"], p["tauB"]
        s.solidPhi = (p["SolidRhoR"] - p["SolidRhoB"]) / (p["SolidRhoR"] + p["SolidRhoB"])
        s.vzR, s.vzB, s.rhoOutR, s.rhoOutB = p["velocityZR"], p["velocityZB"], p["densityRL"], p["densityBL"]
        s.mrt = 1 if p["relax"] == "MRT" else 0
        s.rcA, s.rcD = float(p.get("recolor_axis", 0.0)), float(p.get("recolor_diag", 0.0))
        s.inletP, s.rhoInR, s.rhoInB = int(p["inlet"] == "Dirichlet"), float(p["densityRH"]), float(p["densityBH"])
        s.conv = int(p["outlet"] == "Convective")
        self._names = ("fR", "fB", "gR", "gB", "rhoR", "rhoB", "phi", "vx", "vy", "vz", "Gx", "Gy", "Gz")
        for name in self._names:
            setattr(s, name, getattr(self, "_" + name).ctypes.data_as(F64P))
        self._s, self._L = s, L
        a = np.ascontiguousarray(rhoR0, dtype=np.float64); b = np.ascontiguousarray(rhoB0, dtype=np.float64)
        L.rk3d_init(C.byref(s), a.ctypes.data_as(F64P), b.ctypes.data_as(F64P))

    def set_populations(self, fR, fB):
        """start from given populations of the streamed lattice, [nz][ny][nx][19] per colour (zeros off the fluid), instead of
        w rho at rest: the counterpart of lbmpm_rk3d_set_pdf(..., post_collision = 0) / set_macro for the tests"""
        n = int(np.prod(self.shape)) * 19
        fluid = (self.dom.reshape(-1) == 1)[:, None]
        for name, a in (("fR", fR), ("fB", fB)):
            dst = np.ctypeslib.as_array(getattr(self._s, name), shape=(n,)).reshape(-1, 19)
            dst[:] = np.where(fluid, np.asarray(a, dtype=np.float64).reshape(-1, 19), 0.0)
        return self

    def run(self, n):
        self._L.rk3d_run(C.byref(self._s), C.c_int64(int(n)))
        return self

    def macro(self):
        """boundary planes + rho, u, phi of the current populations (start-of-step view).
        NOTE: modifies the boundary planes exactly like the first half of a step does, which is
        idempotent with respect to the following run()."""
        self._L.rk3d_bc_and_macro_public(C.byref(self._s))
        return self

    def field(self, name):
        # populations swap buffers inside the C struct; read through the struct pointers
        ptr = getattr(self._s, name)
        n = int(np.prod(self.shape)) * (19 if name in ("fR", "fB") else 1)
        a = np.ctypeslib.as_array(ptr, shape=(n,)).copy()
        return a.reshape(self.shape + ((19,) if name in ("fR", "fB") else ()))
