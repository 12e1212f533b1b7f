"""ctypes front-end of oracle/sc_oracle.c (Shan-Chen / explicit-forcing D2Q9, reference layout).

TEST INFRASTRUCTURE ONLY -- see oracle/__init__.py.
"""
import ctypes as C

import numpy as np

from . import lib
from .rk import simple_geometry, image_geometry  # noqa: F401  (same geometry rules)

I64P = C.POINTER(C.c_int64)
F64P = C.POINTER(C.c_double)
U8P = C.POINTER(C.c_uint8)


def _p(a, t):
    return a.ctypes.data_as(t)


def transformation_matrix():
    """BasicD2Q9.transformationMatrix, SimpleD2Q9.py:107-124."""
    M = np.zeros((9, 9))
    M[0, :] = 1.; M[1, 0] = -4.; M[1, 1:5] = -1.; M[1, 5:] = 2.
    M[2, 0] = 4.; M[2, 1:5] = -2.; M[2, 5:] = 1.
    M[3, 1] = 1.; M[3, 3] = -1.; M[3, 5] = 1.; M[3, 6:8] = -1.; M[3, -1] = 1.
    M[4, 1] = -2.; M[4, 3] = 2.; M[4, 5] = 1.; M[4, 6:8] = -1.; M[4, 8] = 1.
    M[5, 2] = 1.; M[5, 4] = -1.; M[5, 5:7] = 1.; M[5, 7:] = -1.
    M[6, 2] = -2.; M[6, 4] = 2.; M[6, 5:7] = 1.; M[6, 7:] = -1.
    M[7, 1] = 1.; M[7, 2] = -1.; M[7, 3] = 1.; M[7, 4] = -1.
    M[8, 5] = 1.; M[8, -3] = -1.; M[8, -2] = 1.; M[8, -1] = -1.
    return M


def collision_matrices(tau):
    """Lambda_k = M^-1 S_k M with S_k = diag(1,.6,1.5,1,1.2,1,1.2,1/tau_k,1/tau_k)
    (ShanChenD2Q9.py:96-106, :484-496; inverse by scipy.linalg.inv there)."""
    import scipy.linalg as slin
    M = transformation_matrix()
    Minv = slin.inv(M)
    out = np.empty((2, 9, 9))
    for k in range(2):
        S = np.zeros((9, 9))
        d = np.ones(9); d[1] = 0.6; d[2] = 1.5; d[0] = 1.; d[4] = 1.2; d[6] = 1.2
        for i in range(9):
            S[i, i] = d[i]
            if i in (7, 8):
                S[i, i] = 1. / tau[k]
        out[k] = np.dot(np.dot(Minv, S), M)
    return out


class _Sim(C.Structure):
    _fields_ = [("N", C.c_int64), ("nx", C.c_int64), ("ny", C.c_int64),
                ("fluidNodes", I64P), ("nbr", I64P),
                ("tau", C.c_double * 2), ("G", C.c_double * 4), ("Gs", C.c_double * 2),
                ("vyIn", C.c_double * 2), ("mrt", C.c_int), ("outletType", C.c_int), ("Lam", F64P)] + \
               [(n, F64P) for n in ("f", "fOld", "fNew", "rho", "psi", "Fx", "Fy", "ux", "uy", "feq",
                                    "ff", "fM", "ffM", "vx", "vy")] + \
               [("scheme", C.c_int), ("nbrX", I64P), ("wX", C.c_double * 36), ("inletMethod", C.c_int), ("fOldValid", C.c_int)]


DEFAULT_PARAMS = dict(inter="EFS", relax="SRT", rho0=1.0, rho1=1.0, bg0=0.02, bg1=0.02, tau0=1.0, tau1=1.0,
                      G=0.20, Gs0=-0.14, Gs1=0.14, outlet="Dirichlet", method="ZouHe", vy0=0.0, vy1=-5.03e-4, scheme=4)


def initial_densities(dom, image, p):
    """ShanChenD2Q9.py:746-767 (no image: fluid 0 below row ny-10) and :768-787 (image: ny-20)."""
    ny, nx = dom.shape
    ii = np.mgrid[0:ny, 0:nx][0]
    lower = ii < (ny - 20 if image else ny - 10)
    fluid = dom == 1
    r = np.zeros((2, ny, nx))
    r[0][fluid & lower] = p["rho0"]; r[1][fluid & lower] = p["bg1"]
    r[1][fluid & ~lower] = p["rho1"]; r[0][fluid & ~lower] = p["bg0"]
    return r


class SCOracle:
    """runOptimizedEFLBM (inter='EFS') or runOptimizedLBM (inter='ShanChen') on the CPU."""

    def __init__(self, dom, params=None, rho_init=None, image=False):
        L = lib()
        p = dict(DEFAULT_PARAMS); p.update(params or {})
        self.p = p
        dom = np.ascontiguousarray(dom, dtype=np.uint8)
        ny, nx = dom.shape
        self.nx, self.ny, self.dom = nx, ny, dom
        fluid = np.empty(nx * ny, np.int64); newidx = np.empty(nx * ny, np.int64)
        L.sc_compact.restype = C.c_int64
        N = L.sc_compact(C.c_int64(nx), C.c_int64(ny), _p(dom, U8P), _p(fluid, I64P), _p(newidx, I64P))
        self.N = N
        self.fluidNodes = fluid[:N].copy()
        self.nbr = np.empty(8 * N, np.int64)
        L.rk_fill_neighbors(C.c_int64(N), C.c_int64(nx), C.c_int64(ny), _p(self.fluidNodes, I64P),
                            _p(newidx, I64P), _p(self.nbr, I64P))
        if rho_init is None:
            rho_init = initial_densities(dom, image, p)
        sel = dom.reshape(-1) == 1
        self.rho = np.ascontiguousarray(rho_init.reshape(2, -1)[:, sel])
        w = np.array([4. / 9.] + [1. / 9.] * 4 + [1. / 36.] * 4)
        self.f = np.ascontiguousarray(w[None, None, :] * self.rho[:, :, None])   # weightsCoeff * rho
        z = lambda *s: np.zeros(s)
        for name in ("fOld", "fNew", "feq", "ff", "fM", "ffM"):
            setattr(self, name, z(2, N, 9))
        for name in ("psi", "Fx", "Fy"):
            setattr(self, name, z(2, N))
        for name in ("ux", "uy", "vx", "vy"):
            setattr(self, name, z(N))
        self.efs = p["inter"] == "EFS"
        self.tau = np.array([p["tau0"], p["tau1"]])
        self.Lam = collision_matrices(self.tau) if p["relax"] == "MRT" else np.zeros((2, 9, 9))
        s = _Sim()
        s.N, s.nx, s.ny = N, nx, ny
        s.fluidNodes, s.nbr = _p(self.fluidNodes, I64P), _p(self.nbr, I64P)
        s.tau[0], s.tau[1] = p["tau0"], p["tau1"]
        s.G[0], s.G[1], s.G[2], s.G[3] = 0.0, p["G"], p["G"], 0.0     # interCoeff, D:375-383
        s.Gs[0], s.Gs[1] = p["Gs0"], p["Gs1"]
        s.vyIn[0], s.vyIn[1] = p["vy0"], p["vy1"]
        s.mrt = 1 if p["relax"] == "MRT" else 0
        s.outletType = {"Dirichlet": 0, "Convective": 1, "Periodic": 2, "Freeflow": 3}[p["outlet"]]
        if p["outlet"] == "Freeflow" and (p["relax"] != "SRT" or not self.efs):
            raise ValueError("'Freeflow' is restated for the explicit forcing loop with SRT (with MRT the reference run turns NaN)")
        s.inletMethod = 1 if p["method"] == "Chang" else 0
        s.Lam = _p(self.Lam, F64P)
        for name in ("f", "fOld", "fNew", "rho", "psi", "Fx", "Fy", "ux", "uy", "feq", "ff", "fM", "ffM",
                     "vx", "vy"):
            setattr(s, name, _p(getattr(self, name), F64P))
        s.scheme = int(p["scheme"])
        if s.scheme not in (4, 8, 10):
            raise ValueError("ExplicitScheme must be 4, 8 or 10")
        if s.scheme != 4:
            nn = 24 if s.scheme == 8 else 36
            self.nbrX = np.empty(nn * N, np.int64)
            L.sc_fill_neighbors_iso(C.c_int64(N), C.c_int64(nx), C.c_int64(ny), C.c_int(nn), _p(self.fluidNodes, I64P),
                                    _p(newidx, I64P), _p(self.nbrX, I64P))
            s.nbrX = _p(self.nbrX, I64P)
            L.sc_iso_weights(C.c_int(s.scheme), s.wX)
        self._s, self._L = s, L
        self.iterations = 0
        if self.efs:
            L.sc_efs_prepare(C.byref(s))

    def run(self, n):
        if self.efs:
            self._L.sc_efs_run(C.byref(self._s), C.c_int64(int(n)))
        else:
            self._L.sc_sc_run(C.byref(self._s), C.c_int64(int(n)))
        self.iterations += int(n)
        return self

    def threads(self):
        return int(self._L.rk_oracle_threads())

    def dense(self, name):
        a = getattr(self, name)
        lead = a.shape[:-1] if a.ndim == 1 else ()
        if a.ndim == 1:
            out = np.zeros(self.ny * self.nx); out[self.fluidNodes] = a
            return out.reshape(self.ny, self.nx)
        if a.ndim == 2:
            out = np.zeros((2, self.ny * self.nx)); out[:, self.fluidNodes] = a
            return out.reshape(2, self.ny, self.nx)
        out = np.zeros((2, self.ny * self.nx, 9)); out[:, self.fluidNodes, :] = a
        return out.reshape(2, self.ny, self.nx, 9)
