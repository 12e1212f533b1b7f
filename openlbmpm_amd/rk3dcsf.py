"""Colour-gradient D3Q19 solver with continuum-surface-force tension -- Python face of lbmpm_rk3dcsf_* (include/lbmpm.h).

[SurfaceTension] SurfaceTensionType = 'CSF' in three dimensions: the reference's 2-D CSF loop (RKColorGradientLBM.runRKColorGradient2DCSF,
RKCG2D/RKD2Q9.py:1295-1490) carried to D3Q19 with z as the flow axis.  All arithmetic happens in liblbmpm_hip.so.
"""
import ctypes as C

import numpy as np

from . import _lib
from ._lib import F64P, U8P, RK3DCSFConfig, check

FIELDS = dict(fR=0, fB=1, rhoR=2, rhoB=3, vx=4, vy=5, vz=6, phi=7, Gx=8, Gy=9, Gz=10, Fx=11, Fy=12, Fz=13, K=14, nsx=15, nsy=16, nsz=17,
              kind=18, rec_fR=30, rec_fB=31, rec_rhoR=32, rec_rhoB=33, rec_vx=34, rec_vy=35, rec_vz=36, rec_phi=37)
_PDF_FIELDS = {"fR", "fB", "rec_fR", "rec_fB"}

# the 2-D ini's parameters (IniFiles/RKtwophasesetup2D.ini) under the 3-D ini's key names for the flow axis (RKtwophasesetup3D.ini:27-38)
DEFAULT_PARAMS = dict(sigma=0.1, theta=60.0, wetting=2, beta=0.7, delta=0.98, tauR=1.0, tauB=1.0, tautype=2, relax="MRT",
                      inlet="Neumann", outlet="Dirichlet", velocityZR=-1.0e-4, velocityZB=0.0, densityBH=5e-8, densityRH=1.00536,
                      densityBL=1.0, densityRL=5e-8, rates=None, variant=0, bulk_epsilon=0.0)


def _f64(a):
    return np.ascontiguousarray(a, dtype=np.float64)


class RK3DCSFSolver:
    def __init__(self, is_domain, params=None, device=0, diagnostics=False):
        L = _lib.lib()
        p = dict(DEFAULT_PARAMS)
        p.update(params or {})
        unknown = set(p) - set(DEFAULT_PARAMS)
        if unknown:
            raise KeyError("unknown RK3DCSF parameters: %s" % sorted(unknown))
        self.params = p
        dom = np.ascontiguousarray(is_domain, dtype=np.uint8)
        if dom.ndim != 3:
            raise TypeError("is_domain must be a 3-D array [nz, ny, nx]")
        self.nz, self.ny, self.nx = dom.shape
        self.shape = dom.shape
        self.is_domain = dom
        cfg = RK3DCSFConfig()
        cfg.nx, cfg.ny, cfg.nz = self.nx, self.ny, self.nz
        cfg.surface_tension, cfg.contact_angle_deg = p["sigma"], p["theta"]
        cfg.beta, cfg.delta, cfg.tau_r, cfg.tau_b = p["beta"], p["delta"], p["tauR"], p["tauB"]
        cfg.inlet_velocity_z = p["velocityZB"] + p["velocityZR"]
        cfg.inlet_rho_r, cfg.inlet_rho_b = p["densityRH"], p["densityBH"]
        cfg.outlet_rho_total = p["densityBL"] + p["densityRL"]
        cfg.wetting_type, cfg.tau_type = int(p["wetting"]), int(p["tautype"])
        if p["relax"] not in ("SRT", "MRT"):
            raise ValueError("RelaxationType must be 'SRT' or 'MRT'")
        cfg.relaxation = 1 if p["relax"] == "MRT" else 0
        if p["inlet"] not in ("Neumann", "Dirichlet"):
            raise ValueError("BoundaryTypeInlet must be 'Neumann' or 'Dirichlet'")
        if p["outlet"] not in ("Dirichlet", "Convective"):
            raise ValueError("BoundaryTypeOutlet must be 'Dirichlet' or 'Convective'")
        cfg.inlet_type = 0 if p["inlet"] == "Neumann" else 1
        cfg.outlet_type = 0 if p["outlet"] == "Dirichlet" else 1
        cfg.device = int(device)
        cfg.variant = int(p["variant"])        # 1: no bulk skip (cross-check)
        cfg.bulk_epsilon = float(p["bulk_epsilon"])      # 0: exact (2^-51); opt-in: cut a colour's tail below this fraction of the density
        if p["rates"] is not None:
            if len(p["rates"]) != 6:
                raise ValueError("rates = (s_e, s_eps, s_q, s_pi, s_m, rate of the conserved moments)")
            for i, r in enumerate(p["rates"]):
                cfg.mrt_rates[i] = float(r)
        self._h = C.c_void_p()
        check(L.lbmpm_rk3dcsf_create(C.byref(cfg), dom.ctypes.data_as(U8P), C.byref(self._h)), "lbmpm_rk3dcsf_create")
        self._L = L
        if diagnostics:
            self.enable_diagnostics(True)

    def close(self):
        if getattr(self, "_h", None) is not None and self._h:
            self._L.lbmpm_rk3dcsf_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _ptr(self, a, shape, what):
        if a is None:
            return None
        a = _f64(a)
        if a.shape != shape:
            raise TypeError("%s must have shape %s" % (what, shape))
        self._keep.append(a)
        return a.ctypes.data_as(F64P)

    def set_macro(self, rhoR, rhoB, vx=None, vy=None, vz=None):
        self._keep = []
        ptr = [self._ptr(a, self.shape, "macroscopic arrays") for a in (rhoR, rhoB, vx, vy, vz)]
        check(self._L.lbmpm_rk3dcsf_set_macro(self._h, *ptr), "lbmpm_rk3dcsf_set_macro")
        self._keep = []

    def set_pdf(self, fR, fB, force=None):
        """streamed populations [nz][ny][nx][19] per colour + the force of the last step (Fx, Fy, Fz): the restart"""
        self._keep = []
        ptr = [self._ptr(a, self.shape + (19,), "populations") for a in (fR, fB)]
        ptr += [self._ptr(a, self.shape, "force") for a in (force or (None, None, None))]
        check(self._L.lbmpm_rk3dcsf_set_pdf(self._h, *ptr), "lbmpm_rk3dcsf_set_pdf")
        self._keep = []

    def enable_diagnostics(self, on=True):
        check(self._L.lbmpm_rk3dcsf_enable_diagnostics(self._h, 1 if on else 0), "enable_diagnostics")

    def step(self, nsteps=1):
        check(self._L.lbmpm_rk3dcsf_step(self._h, int(nsteps)), "lbmpm_rk3dcsf_step")

    def step_timed(self, nsteps):
        """(ms_total, ms of the csf3d_collide launches) by HIP events on the solver's stream"""
        a, b = C.c_double(0), C.c_double(0)
        check(self._L.lbmpm_rk3dcsf_step_timed(self._h, int(nsteps), C.byref(a), C.byref(b)), "lbmpm_rk3dcsf_step_timed")
        return a.value, b.value

    def sync(self):
        check(self._L.lbmpm_rk3dcsf_sync(self._h), "lbmpm_rk3dcsf_sync")

    def get(self, name):
        out = np.empty(self.shape + ((19,) if name in _PDF_FIELDS else ()), dtype=np.float64)
        check(self._L.lbmpm_rk3dcsf_get_field(self._h, FIELDS[name], out.ctypes.data_as(F64P)), "get_field(%s)" % name)
        return out

    @property
    def num_fluid_nodes(self):
        return int(self._L.lbmpm_rk3dcsf_num_fluid_nodes(self._h))

    @property
    def num_wetting_solids(self):
        return int(self._L.lbmpm_rk3dcsf_num_wetting_solids(self._h))

    @property
    def bulk_cells(self):
        """fluid cells whose block of 256 skipped the phase-field pull, the gradient and the curvature in the last step"""
        return int(self._L.lbmpm_rk3dcsf_bulk_cells(self._h))

    @property
    def steps_done(self):
        return int(self._L.lbmpm_rk3dcsf_steps_done(self._h))

    @property
    def device_bytes(self):
        return int(self._L.lbmpm_rk3dcsf_device_bytes(self._h))

    @property
    def dominant_kernel(self):
        return self._L.lbmpm_rk3dcsf_dominant_kernel(self._h).decode()
