"""Fused two-component Shan-Chen D2Q9 solver (original Shan-Chen and explicit forcing, SRT/MRT)
-- Python face of lbmpm_sc2d_* (include/lbmpm.h).

Replaces the per-kernel loops of ShanChenD2Q9.runOptimizedLBM / runOptimizedEFLBM (reference
ShanChen2D/ShanChenD2Q9.py:1433-1629, :1631-2087).  All arithmetic happens in liblbmpm_hip.so.
"""
import ctypes as C

import numpy as np

from . import _lib
from ._lib import F64P, U8P, SC2DConfig, check

FIELDS = dict(f0=0, f1=1, rho0=2, rho1=3, vx=4, vy=5, Fx0=6, Fx1=7, Fy0=8, Fy1=9, ueqx=10, ueqy=11,
              rec_f0=20, rec_f1=21, rec_rho0=22, rec_rho1=23)

# parameter names follow the reference ini files (twophasesetup.ini, efs2D.ini / shanchen2D.ini)
DEFAULT_PARAMS = dict(inter="EFS", relax="SRT", tau0=1.0, tau1=1.0, G=0.20, Gs0=-0.14, Gs1=0.14,
                      outlet="Dirichlet", method="ZouHe", vy0=0.0, vy1=-5.03e-4, scheme=4)


def _f64(a):
    return np.ascontiguousarray(a, dtype=np.float64)


class SC2DSolver:
    def __init__(self, is_domain, params=None, device=0, diagnostics=False):
        L = _lib.lib()
        p = dict(DEFAULT_PARAMS); p.update(params or {})
        unknown = set(p) - set(DEFAULT_PARAMS)
        if unknown:
            raise KeyError("unknown SC2D parameters: %s" % sorted(unknown))
        self.params = p
        dom = np.ascontiguousarray(is_domain, dtype=np.uint8)
        if dom.ndim != 2:
            raise TypeError("is_domain must be a 2-D array [ny, nx]")
        self.ny, self.nx = dom.shape
        self.is_domain = dom
        if p["inter"] not in ("EFS", "ShanChen"):
            raise ValueError("InteractionType must be 'ShanChen' or 'EFS'")
        if p["relax"] not in ("SRT", "MRT"):
            raise ValueError("RelaxationType must be 'SRT' or 'MRT' ('TRT' is a stub in the reference)")
        if p["outlet"] not in ("Dirichlet", "Convective", "Periodic", "Freeflow"):
            raise ValueError("BoundaryTypeOutlet must be 'Dirichlet', 'Convective' or 'Freeflow'")
        if p["method"] not in ("ZouHe", "Chang"):
            raise ValueError("BoundaryMethod must be 'ZouHe' or 'Chang'")
        cfg = SC2DConfig()
        cfg.nx, cfg.ny = self.nx, self.ny
        cfg.model = 1 if p["inter"] == "EFS" else 0
        cfg.relaxation = 1 if p["relax"] == "MRT" else 0
        cfg.tau[0], cfg.tau[1] = p["tau0"], p["tau1"]
        cfg.g_fluid = p["G"]
        cfg.g_solid[0], cfg.g_solid[1] = p["Gs0"], p["Gs1"]
        cfg.outlet_type = {"Dirichlet": 0, "Convective": 1, "Periodic": 2, "Freeflow": 3}[p["outlet"]]     # Periodic: no boundary kernels (API only)
        cfg.inlet_method = 1 if p["method"] == "Chang" else 0
        cfg.inlet_velocity_y[0], cfg.inlet_velocity_y[1] = p["vy0"], p["vy1"]
        cfg.device = int(device); cfg.variant = 0
        cfg.force_scheme = int(p["scheme"])
        self._h = C.c_void_p()
        check(L.lbmpm_sc2d_create(C.byref(cfg), dom.ctypes.data_as(U8P), C.byref(self._h)), "lbmpm_sc2d_create")
        self._L = L
        if diagnostics:
            self.enable_diagnostics(True)

    def close(self):
        if getattr(self, "_h", None) is not None and self._h:
            self._L.lbmpm_sc2d_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def set_pdf(self, f0, f1):
        f0, f1 = _f64(f0), _f64(f1)
        shape = (self.ny, self.nx, 9)
        if f0.shape != shape or f1.shape != shape:
            raise TypeError("pdf arrays must have shape %s" % (shape,))
        check(self._L.lbmpm_sc2d_set_pdf(self._h, f0.ctypes.data_as(F64P), f1.ctypes.data_as(F64P)), "set_pdf")

    def set_density(self, rho0, rho1):
        rho0, rho1 = _f64(rho0), _f64(rho1)
        if rho0.shape != (self.ny, self.nx) or rho1.shape != (self.ny, self.nx):
            raise TypeError("density arrays must have shape %s" % ((self.ny, self.nx),))
        check(self._L.lbmpm_sc2d_set_density(self._h, rho0.ctypes.data_as(F64P), rho1.ctypes.data_as(F64P)),
              "set_density")

    def enable_diagnostics(self, on=True):
        check(self._L.lbmpm_sc2d_enable_diagnostics(self._h, 1 if on else 0), "enable_diagnostics")

    def step(self, nsteps=1):
        check(self._L.lbmpm_sc2d_step(self._h, int(nsteps)), "lbmpm_sc2d_step")

    def step_timed(self, nsteps):
        a, b = C.c_double(0), C.c_double(0)
        check(self._L.lbmpm_sc2d_step_timed(self._h, int(nsteps), C.byref(a), C.byref(b)), "step_timed")
        return a.value, b.value

    def sync(self):
        check(self._L.lbmpm_sc2d_sync(self._h), "sync")

    def get(self, name):
        shape = (self.ny, self.nx, 9) if name in ("f0", "f1", "rec_f0", "rec_f1") else (self.ny, self.nx)
        out = np.empty(shape, dtype=np.float64)
        check(self._L.lbmpm_sc2d_get_field(self._h, FIELDS[name], out.ctypes.data_as(F64P)), "get_field(%s)" % name)
        return out

    def get_compact(self, name):
        a = self.get(name)
        sel = self.is_domain.reshape(-1) == 1
        return a.reshape((self.ny * self.nx,) + a.shape[2:])[sel]

    @property
    def num_fluid_nodes(self):
        return int(self._L.lbmpm_sc2d_num_fluid_nodes(self._h))

    @property
    def device_bytes(self):
        """device memory held by this context"""
        return int(self._L.lbmpm_sc2d_device_bytes(self._h))

    @property
    def steps_done(self):
        return int(self._L.lbmpm_sc2d_steps_done(self._h))

    @property
    def dominant_kernel(self):
        return self._L.lbmpm_sc2d_dominant_kernel(self._h).decode()
