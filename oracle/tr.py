"""ctypes front-end of oracle/tr_oracle.c (D2Q5 tracer transport) and the coupled colour-gradient +
tracer loop.  TEST INFRASTRUCTURE ONLY -- see oracle/__init__.py."""
import ctypes as C

import numpy as np

from . import lib
from .rk import RKOracle

I64P = C.POINTER(C.c_int64)
F64P = C.POINTER(C.c_double)


def _p(a, t):
    return a.ctypes.data_as(t)


def tracer_matrices(diffX, diffY, dXY, dYX):
    """Transport2DRK.py:313-347: D2Q5 weights, M_T, S_T and the collision matrix -M^-1 S^-1."""
    import scipy.linalg as slin
    nT = len(diffX)
    M = np.ones([5, 5])
    M[1, 0] = 0; M[1, 2] = -1.; M[1, 3:] = 0.
    M[2, :3] = 0.; M[2, 4] = -1.
    M[3, 0] = 4.; M[3, 1:] = -1.
    M[4, 0] = 0.; M[4, 3:] = -1.
    Minv = slin.inv(M)
    A = np.zeros([nT, 5, 5])
    for i in range(nT):
        S = np.zeros([5, 5])
        S[1, 1] = (0.5 + 3. * diffX[i]); S[2, 2] = (0.5 + 3. * diffY[i])
        S[1, 2] = 3. * dXY; S[2, 1] = 3. * dYX
        S[0, 0] = 1.0; S[3, 3] = 1.0; S[4, 4] = 1.0
        A[i] = -np.dot(Minv, slin.inv(S))
    return M, A


class _Tr(C.Structure):
    _fields_ = [("N", C.c_int64), ("nx", C.c_int64), ("ny", C.c_int64), ("nT", C.c_int), ("freeOutlet", C.c_int),
                ("dirichletInlet", C.c_int), ("fluidNodes", I64P), ("nbr4", I64P), ("M", F64P), ("A", F64P),
                ("beta", F64P), ("cb", F64P), ("crit", C.c_double), ("g", F64P), ("gNew", F64P), ("C", F64P),
                ("ind", F64P), ("reaction", C.c_int), ("rate", C.c_double * 1), ("J", C.c_double * 20)]


DEFAULT_TRACER = dict(diffX=(1. / 6.,), diffY=(1. / 6.,), dXY=0.0, dYX=0.0, beta=(1.0,), crit=0.5,
                      inlet_conc=(1.0,), free_outlet=True, dirichlet_inlet=True, reaction_rate=0.0, diffJ=None)


class CoupledOracle:
    """Colour-gradient CSF flow (oracle/rk_oracle.c) with the tracer sub-step spliced in where
    Transport2DRK.runTransport2DMPMCRKNew does it (after the wetting-corrected gradient,
    Transport2DRK.py:1316-1418; before the CSF force, :1435)."""

    def __init__(self, dom, flow_params, rhoR0, rhoB0, conc0, tracer=None):
        self.flow = RKOracle(dom, flow_params, rhoR0, rhoB0)
        L = lib()
        t = dict(DEFAULT_TRACER); t.update(tracer or {})
        f = self.flow
        N = f.N
        nT = len(t["diffX"])
        self.nT = nT
        newidx = -np.ones(f.nx * f.ny, dtype=np.int64)
        newidx[f.fluidNodes] = np.arange(N)
        self.nbr4 = np.empty(4 * N, np.int64)
        L.tr_fill_neighbors(C.c_int64(N), C.c_int64(f.nx), C.c_int64(f.ny), _p(f.fluidNodes, I64P), _p(newidx, I64P),
                            _p(self.nbr4, I64P))
        self.M, self.A = tracer_matrices(t["diffX"], t["diffY"], t["dXY"], t["dYX"])
        self.beta = np.array(t["beta"], dtype=np.float64); self.cb = np.array(t["inlet_conc"], dtype=np.float64)
        sel = f.dom.reshape(-1) == 1
        self.C = np.ascontiguousarray(np.asarray(conc0, dtype=np.float64).reshape(nT, -1)[:, sel])
        w5 = np.array([1. / 3.] + [1. / 6.] * 4)
        self.g = np.ascontiguousarray(self.C[:, :, None] * w5[None, None, :])
        self.gNew = np.zeros_like(self.g); self.ind = np.zeros(N)
        s = _Tr()
        s.N, s.nx, s.ny, s.nT = N, f.nx, f.ny, nT
        s.freeOutlet, s.dirichletInlet = int(t["free_outlet"]), int(t["dirichlet_inlet"])
        s.fluidNodes, s.nbr4 = _p(f.fluidNodes, I64P), _p(self.nbr4, I64P)
        s.M, s.A, s.beta, s.cb = _p(self.M, F64P), _p(self.A, F64P), _p(self.beta, F64P), _p(self.cb, F64P)
        s.crit = t["crit"]
        s.g, s.gNew, s.C, s.ind = _p(self.g, F64P), _p(self.gNew, F64P), _p(self.C, F64P), _p(self.ind, F64P)
        s.reaction = 1 if t["reaction_rate"] else 0
        if s.reaction:
            if nT != 3:
                raise ValueError("the reaction couples exactly three tracers")
            s.rate[0] = t["reaction_rate"]
            dj = t["diffJ"] or (1. / 3.,) * nT
            for i in range(nT):
                for j in range(5):
                    s.J[5 * i + j] = dj[i] if j == 0 else (1. - dj[i]) / 4.
        self._s, self._L = s, L

    def run(self, n):
        f, L = self.flow, self._L
        for _ in range(int(n)):
            L.rk_csf_step_a_transport(C.byref(f._s))        # boundary rows first, densities summed afterwards (Transport2DRK.py:1199-1287)
            L.tr_substep(C.byref(self._s), _p(f.rhoR, F64P), _p(f.vx, F64P), _p(f.vy, F64P), _p(f.Gx, F64P), _p(f.Gy, F64P))
            L.rk_csf_step_b(C.byref(f._s))
        return self
