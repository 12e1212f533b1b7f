"""Fused colour-gradient D2Q9 (CSF) solver -- Python face of lbmpm_rk2d_* (include/lbmpm.h).

Replaces the per-kernel time loop of RKColorGradientLBM.runRKColorGradient2DCSF
(reference RKCG2D/RKD2Q9.py:1295-1490).  All arithmetic happens in liblbmpm_hip.so.
"""
import ctypes as C

import numpy as np

from . import _lib
from ._lib import F64P, U8P, RK2DConfig, RK2DPerturbation, TracerConfig, check

FIELDS = dict(fR=0, fB=1, rhoR=2, rhoB=3, vx=4, vy=5, phi=6, Gx=7, Gy=8, Fx=9, Fy=10, K=11,
              rec_fR=20, rec_fB=21, rec_rhoR=22, rec_rhoB=23, rec_vx=24, rec_vy=25)
_PDF_FIELDS = {"fR", "fB", "rec_fR", "rec_fB"}

# parameter names follow the reference ini (IniFiles/RKtwophasesetup2D.ini)
DEFAULT_PARAMS = dict(sigma=0.1, theta=60.0, wetting=2, beta=0.7, delta=0.98, tauR=1.0, tauB=1.0,
                      tautype=2, relax="MRT", inlet="Neumann", outlet="Dirichlet",
                      vyR=-1.0e-4, vyB=0.0, rhoBH=5e-8, rhoRH=1.00536, rhoBL=1.0, rhoRL=5e-8)


def _f64(a):
    return np.ascontiguousarray(a, dtype=np.float64)


class RK2DSolver:
    def __init__(self, is_domain, params=None, device=0, variant=0, diagnostics=False, perturbation=None):
        """perturbation = dict(AkR=, AkB=, solidPhi=): the step of SurfaceTensionType 'Perturbation' (RKD2Q9.py:978-1223) as one fused
        launch instead of the CSF step (lbmpm_rk2d_set_perturbation): inlet 'Neumann' (vyR / vyB per colour) or 'Dirichlet' (rhoRH /
        rhoBH), outlet 'Dirichlet' (rhoRL / rhoBL per colour) or 'Convective', tauR, tauB, beta, relax of `params`; raises LbmpmError
        (unsupported) for solid nodes in the boundary rows."""
        L = _lib.lib()
        p = dict(DEFAULT_PARAMS)
        p.update(params or {})
        unknown = set(p) - set(DEFAULT_PARAMS)
        if unknown:
            raise KeyError("unknown RK2D parameters: %s" % sorted(unknown))
        self.params = p
        dom = np.ascontiguousarray(is_domain, dtype=np.uint8)
        if dom.ndim != 2:
            raise TypeError("is_domain must be a 2-D array [ny, nx]")
        self.ny, self.nx = dom.shape
        self.is_domain = dom
        cfg = RK2DConfig()
        cfg.nx, cfg.ny = self.nx, self.ny
        cfg.surface_tension = p["sigma"]; cfg.contact_angle_deg = p["theta"]
        cfg.wetting_type = int(p["wetting"])
        cfg.beta = p["beta"]; cfg.delta = p["delta"]
        cfg.tau_r = p["tauR"]; cfg.tau_b = p["tauB"]; cfg.tau_type = int(p["tautype"])
        if p["relax"] not in ("SRT", "MRT"):
            raise ValueError("RelaxationType must be 'SRT' or 'MRT'")
        cfg.relaxation = 1 if p["relax"] == "MRT" else 0
        if p["inlet"] not in ("Neumann", "Dirichlet"):
            raise ValueError("BoundaryTypeInlet must be 'Neumann' or 'Dirichlet'")
        if p["outlet"] not in ("Dirichlet", "Convective"):
            raise ValueError("BoundaryTypeOutlet must be 'Dirichlet' or 'Convective'")
        cfg.inlet_type = 0 if p["inlet"] == "Neumann" else 1
        cfg.outlet_type = 0 if p["outlet"] == "Dirichlet" else 1
        cfg.inlet_velocity_y = p["vyB"] + p["vyR"]
        cfg.inlet_rho_r = p["rhoRH"]; cfg.inlet_rho_b = p["rhoBH"]
        cfg.outlet_rho_total = p["rhoBL"] + p["rhoRL"]
        cfg.device = int(device); cfg.variant = int(variant)
        self._h = C.c_void_p()
        check(L.lbmpm_rk2d_create(C.byref(cfg), dom.ctypes.data_as(U8P), C.byref(self._h)),
              "lbmpm_rk2d_create")
        self._L = L
        self.model = "CSF"
        if perturbation is not None:
            unknown = set(perturbation) - {"AkR", "AkB", "solidPhi"}
            if unknown:
                raise KeyError("unknown perturbation parameters: %s" % sorted(unknown))
            q = RK2DPerturbation()
            q.ak_r, q.ak_b, q.solid_phi = float(perturbation["AkR"]), float(perturbation["AkB"]), float(perturbation.get("solidPhi", 0.5))
            q.inlet_velocity_y_r, q.inlet_velocity_y_b = p["vyR"], p["vyB"]
            q.outlet_rho_r, q.outlet_rho_b = p["rhoRL"], p["rhoBL"]
            try:
                check(L.lbmpm_rk2d_set_perturbation(self._h, C.byref(q)), "lbmpm_rk2d_set_perturbation")
            except Exception:
                self.close()
                raise
            self.model = "Perturbation"
        if diagnostics:
            self.enable_diagnostics(True)

    # -- life cycle
    def close(self):
        if getattr(self, "_h", None) is not None and self._h:
            self._L.lbmpm_rk2d_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # -- state
    def set_pdf(self, fR, fB):
        fR, fB = _f64(fR), _f64(fB)
        shape = (self.ny, self.nx, 9)
        if fR.shape != shape or fB.shape != shape:
            raise TypeError("pdf arrays must have shape %s" % (shape,))
        check(self._L.lbmpm_rk2d_set_pdf(self._h, fR.ctypes.data_as(F64P), fB.ctypes.data_as(F64P)),
              "lbmpm_rk2d_set_pdf")

    def set_macro(self, rhoR, rhoB, vx=None, vy=None):
        arrs = [_f64(rhoR), _f64(rhoB)] + [None if a is None else _f64(a) for a in (vx, vy)]
        for a in arrs:
            if a is not None and a.shape != (self.ny, self.nx):
                raise TypeError("macroscopic arrays must have shape %s" % ((self.ny, self.nx),))
        ptr = [a.ctypes.data_as(F64P) if a is not None else None for a in arrs]
        check(self._L.lbmpm_rk2d_set_macro(self._h, *ptr), "lbmpm_rk2d_set_macro")

    def enable_diagnostics(self, on=True):
        check(self._L.lbmpm_rk2d_enable_diagnostics(self._h, 1 if on else 0), "enable_diagnostics")

    def set_stream(self, hip_stream_handle):
        check(self._L.lbmpm_rk2d_set_stream(self._h, C.c_void_p(hip_stream_handle)), "set_stream")

    # -- tracer transport (BASELINE config 4)
    def configure_tracers(self, diffX=(1. / 6.,), diffY=(1. / 6.,), dXY=0.0, dYX=0.0, beta=(1.0,), crit=0.5,
                          inlet_conc=(1.0,), free_outlet=True, dirichlet_inlet=True, reaction_rate=0.0, diffJ=None):
        """D2Q5-MRT tracers advected by the flow (keys of the reference's transportsetup.ini,
        Transport2DRK.py:35-311)."""
        n = len(diffX)
        if not (len(diffY) == len(beta) == len(inlet_conc) == n):
            raise ValueError("per-tracer lists must have the same length")
        t = TracerConfig()
        t.num_tracers = n
        for k in range(n):
            t.diffusion_x[k], t.diffusion_y[k] = diffX[k], diffY[k]
            t.beta_interface[k], t.inlet_concentration[k] = beta[k], inlet_conc[k]
        t.diffusion_xy, t.diffusion_yx, t.criteria_rho = dXY, dYX, crit
        t.dirichlet_inlet, t.free_outlet = int(dirichlet_inlet), int(free_outlet)
        t.reaction_rate = float(reaction_rate)           # A + B -> C between tracers 0, 1, 2
        dj = tuple(diffJ) if diffJ is not None else (1. / 3.,) * n
        if len(dj) != n:
            raise ValueError("diffJ needs one value per tracer")
        for k in range(n):
            t.diffusion_j[k] = dj[k]
        check(self._L.lbmpm_rk2d_tracer_configure(self._h, C.byref(t)), "lbmpm_rk2d_tracer_configure")
        self.num_tracers = n

    def set_tracer(self, k, conc):
        a = _f64(conc)
        if a.shape != (self.ny, self.nx):
            raise TypeError("concentration must have shape %s" % ((self.ny, self.nx),))
        check(self._L.lbmpm_rk2d_tracer_set_concentration(self._h, int(k), a.ctypes.data_as(F64P)), "tracer_set")

    def get_tracer(self, k, compact=False):
        out = np.empty((self.ny, self.nx), dtype=np.float64)
        check(self._L.lbmpm_rk2d_tracer_get_concentration(self._h, int(k), out.ctypes.data_as(F64P)), "tracer_get")
        return out.reshape(-1)[self.is_domain.reshape(-1) == 1] if compact else out

    # -- time stepping
    def step(self, nsteps=1):
        check(self._L.lbmpm_rk2d_step(self._h, int(nsteps)), "lbmpm_rk2d_step")

    def step_timed(self, nsteps):
        """returns (ms_total, ms_dominant_kernel) measured with HIP events on the solver stream"""
        a, b = C.c_double(0), C.c_double(0)
        check(self._L.lbmpm_rk2d_step_timed(self._h, int(nsteps), C.byref(a), C.byref(b)),
              "lbmpm_rk2d_step_timed")
        return a.value, b.value

    def sync(self):
        check(self._L.lbmpm_rk2d_sync(self._h), "lbmpm_rk2d_sync")

    # -- results
    def get(self, name):
        fid = FIELDS[name]
        shape = (self.ny, self.nx, 9) if name in _PDF_FIELDS else (self.ny, self.nx)
        out = np.empty(shape, dtype=np.float64)
        check(self._L.lbmpm_rk2d_get_field(self._h, fid, out.ctypes.data_as(F64P)), "get_field(%s)" % name)
        return out

    def get_compact(self, name):
        """Field restricted to fluid nodes in the reference's compaction order (row-major
        scan of isDomain, RKD2Q9.py:668-676)."""
        a = self.get(name)
        sel = self.is_domain.reshape(-1) == 1
        return a.reshape((self.ny * self.nx,) + a.shape[2:])[sel]

    @property
    def num_fluid_nodes(self):
        return int(self._L.lbmpm_rk2d_num_fluid_nodes(self._h))

    @property
    def steps_done(self):
        return int(self._L.lbmpm_rk2d_steps_done(self._h))

    @property
    def device_bytes(self):
        return int(self._L.lbmpm_rk2d_device_bytes(self._h))

    @property
    def dominant_kernel(self):
        return self._L.lbmpm_rk2d_dominant_kernel(self._h).decode()


def mrt_matrices():
    """D2Q9 moment basis (rho, e, eps, j_x, q_x, j_y, q_y, p_xx, p_xy) as a literal, its inverse, and the relaxation rates
    (0, 1.64, 1.54, 0, 1.9, 0, 1.9, *, *) -- the values of RKD2Q9.py:308-340 (slots 7, 8 are set per node to 1/tau).
    The fused solver has them built in; the kernel-level entry points take them as arrays, like the reference's kernels."""
    M = np.array([[1, 1, 1, 1, 1, 1, 1, 1, 1], [-4, -1, -1, -1, -1, 2, 2, 2, 2], [4, -2, -2, -2, -2, 1, 1, 1, 1],
                  [0, 1, 0, -1, 0, 1, -1, -1, 1], [0, -2, 0, 2, 0, 1, -1, -1, 1], [0, 0, 1, 0, -1, 1, 1, -1, -1],
                  [0, 0, -2, 0, 2, 1, 1, -1, -1], [0, 1, -1, 1, -1, 0, 0, 0, 0], [0, 0, 0, 0, 0, 1, -1, 1, -1]], dtype=np.float64)
    S = np.array([0., 1.64, 1.54, 0., 1.9, 0., 1.9, 0., 0.])
    return M, np.linalg.inv(M), S
