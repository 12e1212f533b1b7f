"""ctypes front-end of oracle/rk_oracle.c (colour-gradient D2Q9, reference layout).

TEST INFRASTRUCTURE ONLY -- see oracle/__init__.py.
"""
import ctypes as C
import math

import numpy as np

from . import lib

I64P = C.POINTER(C.c_int64)
F64P = C.POINTER(C.c_double)
U8P = C.POINTER(C.c_uint8)


def _p(a, t):
    return a.ctypes.data_as(t)


def mrt_matrices():
    """M as assembled in RKD2Q9.py:308-336; Minv by numpy.linalg.inv (RKD2Q9.py:337);
    S = (0,1.64,1.54,0,1.9,0,1.9,0,0) (RKD2Q9.py:338-340; slots 7,8 set per node)."""
    M = np.zeros((9, 9))
    M[0, :] = 1.
    M[1, :] = -1.; M[1, 0] = -4.; M[1, 5:] = 2.
    M[2, :] = 1.; M[2, 0] = 4.; M[2, 1:5] = -2.
    M[3, 1] = 1.; M[3, 3] = -1.; M[3, 5] = 1.; M[3, 6:-1] = -1.; M[3, -1] = 1.
    M[4, 1] = -2.; M[4, 3] = 2.; M[4, 5] = 1.; M[4, 6:-1] = -1.; M[4, -1] = 1.
    M[5, 2] = 1.; M[5, 4] = -1.; M[5, 5:7] = 1.; M[5, 7:] = -1.
    M[6, 2] = -2.; M[6, 4] = 2.; M[6, 5:7] = 1.; M[6, 7:] = -1.
    M[7, 1] = 1.; M[7, 2] = -1.; M[7, 3] = 1.; M[7, 4] = -1.
    M[8, 5] = 1.; M[8, 6] = -1.; M[8, 7] = 1.; M[8, 8] = -1.
    Minv = np.linalg.inv(M)
    S = np.zeros(9); S[1] = 1.64; S[2] = 1.54; S[4] = 1.9; S[6] = 1.9
    return M, Minv, S


class _Sim(C.Structure):
    _fields_ = [(n, C.c_int64) for n in ("N", "nx", "ny", "W", "Wf")] + \
               [(n, I64P) for n in ("fluidNodes", "nbr", "nbrWet", "fluidWet")] + \
               [(n, F64P) for n in ("nsx", "nsy")] + \
               [(n, C.c_double) for n in ("sigma", "cosT", "sinT", "beta", "delta", "tauR", "tauB",
                                          "vyIn", "pInB", "pInR", "pOutTotal")] + \
               [(n, C.c_int) for n in ("wettingType", "tauType", "mrt", "inletType", "outletType")] + \
               [(n, F64P) for n in ("M", "Minv", "S")] + \
               [(n, F64P) for n in ("fR", "fB", "fRn", "fBn", "fT", "rhoR", "rhoB", "vx", "vy",
                                    "phi", "phiS", "Gx", "Gy", "Fx", "Fy", "K")]


DEFAULT_PARAMS = dict(sigma=0.1, theta=60.0, wetting=2, beta=0.7, delta=0.98, tauR=1.0, tauB=1.0,
                      tautype=2, relax="MRT", inlet="Neumann", outlet="Dirichlet",
                      vyR=-1.0e-4, vyB=0.0, rhoBH=5e-8, rhoRH=1.00536, rhoBL=1.0, rhoRL=5e-8)


def simple_geometry(nx, ny):
    """ShanChen2D/SimpleGeometry.py:11-27: all void, solid x=0 and x=nx-1 on rows 10:-10."""
    dom = np.ones((ny, nx), dtype=np.uint8)
    dom[10:-10, 0] = 0
    dom[10:-10, -1] = 0
    return dom


def image_geometry(img, nbuf, ratio):
    """RKD2Q9.py:373-414 + 417-443: crop to bounding box of zeros, solid first/last column,
    2*nbuf all-void buffer rows split top(row 0 side)/bottom by ratio."""
    img = np.asarray(img, dtype=np.float64)
    ys, xs = np.nonzero(img == 0.0)
    eff = np.array(img[ys.min():ys.max() + 1, xs.min():xs.max() + 1], copy=True)
    eff[:, 0] = 0.; eff[:, -1] = 0.
    buf = np.full(eff.shape[1], 255.)
    for i in range(2 * nbuf):
        if i < int(2 * nbuf * ratio):
            eff = np.vstack((buf, eff))
        else:
            eff = np.vstack((eff, buf))
    return (eff != 0.0).astype(np.uint8)


def initial_densities(dom, image, nbuf, rho0R=1.0, rho0B=1.0):
    """RKD2Q9.py:459-490 (no image: red disc r<=16 at the centre) and :511-531
    (image: red below the top nbuf rows, blue in them)."""
    ny, nx = dom.shape
    rR = np.zeros((ny, nx)); rB = np.zeros((ny, nx))
    ii, jj = np.mgrid[0:ny, 0:nx]
    if not image:
        cy, cx = int(ny / 2), int(nx / 2)
        red = np.sqrt((ii - cy) * (ii - cy) + (jj - cx) * (jj - cx)) <= 16.
    else:
        red = ii < ny - nbuf
    fluid = dom == 1
    rR[fluid & red] = rho0R
    rB[fluid & ~red] = rho0B
    return rR, rB


class RKOracle:
    """The reference's CSF colour-gradient loop (RKD2Q9.py:1225-1490) on the CPU."""

    def __init__(self, dom, params=None, rhoR0=None, rhoB0=None, image=False, nbuf=10):
        L = lib()
        p = dict(DEFAULT_PARAMS)
        p.update(params or {})
        self.p = p
        dom = np.ascontiguousarray(dom, dtype=np.uint8)
        ny, nx = dom.shape
        self.nx, self.ny, self.dom = nx, ny, dom
        fluid = np.empty(nx * ny, np.int64); newidx = np.empty(nx * ny, np.int64)
        wet = np.empty(nx * ny, np.int64); outN = C.c_int64(0)
        L.rk_compact.restype = C.c_int64
        W = L.rk_compact(C.c_int64(nx), C.c_int64(ny), _p(dom, U8P), _p(fluid, I64P), _p(newidx, I64P),
                         _p(wet, I64P), C.byref(outN))
        N = outN.value
        self.N, self.W = N, W
        self.fluidNodes = fluid[:N].copy(); self.wettingSolidNodes = wet[:W].copy()
        self.newIndex = newidx
        self.nbr = np.empty(8 * N, np.int64)
        L.rk_fill_neighbors(C.c_int64(N), C.c_int64(nx), C.c_int64(ny), _p(self.fluidNodes, I64P),
                            _p(newidx, I64P), _p(self.nbr, I64P))
        self.nbrWet = np.empty(max(8 * W, 1), np.int64)
        if W:
            L.rk_fill_neighbors(C.c_int64(W), C.c_int64(nx), C.c_int64(ny), _p(self.wettingSolidNodes, I64P),
                                _p(newidx, I64P), _p(self.nbrWet, I64P))
        g = np.empty(max(N, 1), np.int64); o = np.empty(max(N, 1), np.int64)
        self.Wf = 0
        if W:
            L.rk_fluid_near_solid.restype = C.c_int64
            self.Wf = L.rk_fluid_near_solid(C.c_int64(nx), C.c_int64(ny), _p(dom, U8P), _p(newidx, I64P),
                                            _p(g, I64P), _p(o, I64P))
        self.fluidWet = g[:self.Wf].copy() if self.Wf else np.zeros(1, np.int64)
        self.fluidWetOriginal = o[:self.Wf].copy() if self.Wf else np.zeros(1, np.int64)
        self.nsx = np.zeros(max(self.Wf, 1)); self.nsy = np.zeros(max(self.Wf, 1))
        if self.Wf:
            L.rk_solid_normals(C.c_int64(nx), C.c_int64(ny), _p(dom, U8P), C.c_int64(self.Wf),
                               _p(self.fluidWetOriginal, I64P), _p(self.nsx, F64P), _p(self.nsy, F64P))
        # initial state: f = rho w (u = 0)  (RKD2Q9.py:577-601)
        if rhoR0 is None:
            rhoR0, rhoB0 = initial_densities(dom, image, nbuf)
        sel = dom.reshape(-1) == 1
        self.rhoR = np.ascontiguousarray(rhoR0.reshape(-1)[sel], dtype=np.float64)
        self.rhoB = np.ascontiguousarray(rhoB0.reshape(-1)[sel], dtype=np.float64)
        w = np.array([4. / 9.] + [1. / 9.] * 4 + [1. / 36.] * 4)
        self.fR = np.ascontiguousarray(self.rhoR[:, None] * w[None, :] * (1 + 0.0))
        self.fB = np.ascontiguousarray(self.rhoB[:, None] * w[None, :] * (1 + 0.0))
        z = lambda *s: np.zeros(s)
        self.fRn, self.fBn = z(N, 9), z(N, 9)
        self.fT = self.fR + self.fB
        self.vx, self.vy, self.phi = z(N), z(N), z(N)
        self.phiS = z(max(W, 1))
        self.Gx, self.Gy, self.Fx, self.Fy = z(N), z(N), z(N), z(N)
        self.K = self.rhoB.copy()          # RKD2Q9.py:1266: KValue starts as a copy of rhoB
        self.M, self.Minv, self.S = mrt_matrices()
        th = p["theta"] / 180. * np.pi
        s = _Sim()
        s.N, s.nx, s.ny, s.W, s.Wf = N, nx, ny, W, self.Wf
        s.fluidNodes, s.nbr = _p(self.fluidNodes, I64P), _p(self.nbr, I64P)
        s.nbrWet, s.fluidWet = _p(self.nbrWet, I64P), _p(self.fluidWet, I64P)
        s.nsx, s.nsy = _p(self.nsx, F64P), _p(self.nsy, F64P)
        s.sigma, s.cosT, s.sinT = p["sigma"], float(np.cos(th)), float(np.sin(th))
        s.beta, s.delta, s.tauR, s.tauB = p["beta"], p["delta"], p["tauR"], p["tauB"]
        s.vyIn = p["vyB"] + p["vyR"]
        s.pInB, s.pInR = p["rhoBH"], p["rhoRH"]
        s.pOutTotal = p["rhoBL"] + p["rhoRL"]
        s.wettingType, s.tauType = int(p["wetting"]), int(p["tautype"])
        s.mrt = 1 if p["relax"] == "MRT" else 0
        s.inletType = 0 if p["inlet"] == "Neumann" else 1
        s.outletType = 0 if p["outlet"] == "Dirichlet" else 1
        s.M, s.Minv, s.S = _p(self.M, F64P), _p(self.Minv, F64P), _p(self.S, F64P)
        for name in ("fR", "fB", "fRn", "fBn", "fT", "rhoR", "rhoB", "vx", "vy", "phi", "phiS",
                     "Gx", "Gy", "Fx", "Fy", "K"):
            setattr(s, name, _p(getattr(self, name), F64P))
        self._s = s
        self._L = L

    def run(self, nsteps):
        self._L.rk_csf_run(C.byref(self._s), C.c_int64(int(nsteps)))
        return self

    def threads(self):
        return int(self._L.rk_oracle_threads())

    def dense(self, name):
        """Scatter a compact field to the dense [ny, nx(,9)] grid (zeros at solid),
        like convertOptTo2D (RKD2Q9.py:902-914)."""
        a = getattr(self, name)
        out = np.zeros((self.ny * self.nx,) + a.shape[1:])
        out[self.fluidNodes] = a
        return out.reshape((self.ny, self.nx) + a.shape[1:])


# ---------------------------------------------------------------- perturbation path (oracle/rk_pert_oracle.c)
class _PertSim(C.Structure):
    _fields_ = [(n, C.c_int64) for n in ("N", "nx", "ny")] + \
               [(n, I64P) for n in ("fluidNodes", "nbr")] + \
               [(n, C.c_double) for n in ("beta", "AkR", "AkB", "solidPhi", "tauR", "tauB", "vyR", "vyB", "pLB", "pLR")] + \
               [("mrt", C.c_int)] + \
               [(n, F64P) for n in ("Bc", "rw", "M", "Minv", "S")] + \
               [(n, F64P) for n in ("fR", "fB", "fRn", "fBn", "fT", "rhoR", "rhoB", "vx", "vy", "phi", "Gx", "Gy")]


PERT_DEFAULTS = dict(beta=1.0, AkR=7.0e-3, AkB=7.0e-3, solidPhi=1.0, tauR=1.0, tauB=1.0, relax="SRT",
                     vyR=0.0, vyB=-1.0e-4, rhoBL=1.0, rhoRL=1.0e-8)
PERT_B = np.array([-2. / 9.] + [1. / 9.] * 4 + [1. / 36.] * 4)      # constantBNew, RKD2Q9.py:131-133


class RKPertOracle:
    """The reference's perturbation colour-gradient loop (RKD2Q9.py:978-1223; velocity inlet A:657 + pressure
    outlet A:1008) with the repairs listed in tests/golden/gen/make_golden_rk_pert.py.  `recolor_weights`
    replaces the w_i of the recolouring term (A:1262-1267) -- None = the reference's."""

    def __init__(self, dom, params=None, fR0=None, fB0=None, rhoR0=None, rhoB0=None, recolor_weights=None):
        L = lib()
        p = dict(PERT_DEFAULTS); p.update(params or {})
        self.p = p
        dom = np.ascontiguousarray(dom, dtype=np.uint8)
        ny, nx = dom.shape
        self.nx, self.ny, self.dom = nx, ny, dom
        self.fluidNodes = np.flatnonzero(dom.reshape(-1) == 1).astype(np.int64)
        N = self.N = int(self.fluidNodes.size)
        newidx = -np.ones(nx * ny, dtype=np.int64); newidx[self.fluidNodes] = np.arange(N)
        self.nbr = np.empty(8 * N, np.int64)
        L.rk_fill_neighbors(C.c_int64(N), C.c_int64(nx), C.c_int64(ny), _p(self.fluidNodes, I64P), _p(newidx, I64P),
                            _p(self.nbr, I64P))
        self.nbr[self.nbr < -1] = -1         # optimizeFluidArray (RKD2Q9.py:603-655) knows fluid / not fluid only
        w = np.array([4. / 9.] + [1. / 9.] * 4 + [1. / 36.] * 4)
        if fR0 is None:
            sel = dom.reshape(-1) == 1
            fR0 = np.asarray(rhoR0, dtype=np.float64).reshape(-1)[sel][:, None] * w[None, :]
            fB0 = np.asarray(rhoB0, dtype=np.float64).reshape(-1)[sel][:, None] * w[None, :]
        self.fR = np.ascontiguousarray(fR0, dtype=np.float64).copy(); self.fB = np.ascontiguousarray(fB0, dtype=np.float64).copy()
        z = lambda *s: np.zeros(s)
        self.fRn, self.fBn, self.fT = z(N, 9), z(N, 9), self.fR + self.fB
        self.rhoR, self.rhoB = self.fR.sum(axis=1), self.fB.sum(axis=1)
        self.vx, self.vy, self.phi, self.Gx, self.Gy = z(N), z(N), z(N), z(N), z(N)
        self.Bc = PERT_B.copy()
        self.rw = None if recolor_weights is None else np.ascontiguousarray(recolor_weights, dtype=np.float64)
        self.M, self.Minv, self.S = mrt_matrices()
        s = _PertSim()
        s.N, s.nx, s.ny = N, nx, ny
        s.fluidNodes, s.nbr = _p(self.fluidNodes, I64P), _p(self.nbr, I64P)
        s.beta, s.AkR, s.AkB, s.solidPhi, s.tauR, s.tauB = p["beta"], p["AkR"], p["AkB"], p["solidPhi"], p["tauR"], p["tauB"]
        s.vyR, s.vyB, s.pLB, s.pLR = p["vyR"], p["vyB"], p["rhoBL"], p["rhoRL"]
        s.mrt = 1 if p["relax"] == "MRT" else 0
        s.Bc = _p(self.Bc, F64P)
        s.rw = _p(self.rw, F64P) if self.rw is not None else C.cast(None, F64P)
        s.M, s.Minv, s.S = _p(self.M, F64P), _p(self.Minv, F64P), _p(self.S, F64P)
        for name in ("fR", "fB", "fRn", "fBn", "fT", "rhoR", "rhoB", "vx", "vy", "phi", "Gx", "Gy"):
            setattr(s, name, _p(getattr(self, name), F64P))
        self._s, self._L = s, L

    def run(self, nsteps, order="repaired"):
        """order: 'repaired' (R3: f_tot summed where the collision kernels need it) or 'literal' (RKD2Q9.py:1065: right after streaming)"""
        (self._L.rk_pert_run if order == "repaired" else self._L.rk_pert_run_literal)(C.byref(self._s), C.c_int64(int(nsteps)))
        return self

    def dense(self, name):
        a = getattr(self, name)
        out = np.zeros((self.ny * self.nx,) + a.shape[1:])
        out[self.fluidNodes] = a
        return out.reshape((self.ny, self.nx) + a.shape[1:])
