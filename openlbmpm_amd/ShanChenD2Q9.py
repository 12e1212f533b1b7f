"""Driver counterpart of the reference's ShanChen2D/ShanChenD2Q9.py (own code, same ini files,
same outputs): class ShanChenD2Q9(pathIniFile).runTypeSCmodel().

Kept: ini grammar (twophasesetup.ini + efs2D.ini / shanchen2D.ini), geometry (SimpleGeometry or
image with 20 buffer rows each side, ShanChenD2Q9.py:544-585), initial densities (:746-787),
dataset names of resultInHDF5 (:940-955) with the reference's hard-coded record cadence (every
80 steps for 'ShanChen' :1561, every 1000 loop iterations for 'EFS' :2029; overridable).
The time loop is lbmpm_sc2d_step.
"""
import os

import numpy as np

from . import config
from .geometry import simple_geometry, image_domain
from .results import RecordGuard, ResultFile
from .sc2d import SC2DSolver


class ShanChenD2Q9:
    def __init__(self, pathIniFile, output_dir=None, image=None, device=0, record_every=None, initial_dir=None,
                 duplicate=None):
        self.path = pathIniFile
        self.initial_dir = initial_dir or os.path.expanduser("~/LBMInitial")
        self.duplicate = duplicate            # (x, y) copies of the pore image, overrides the ini
        self.par = config.read_sc2d(pathIniFile)
        self.output_dir = output_dir or os.path.expanduser("~/LBMResults")
        self.device, self._image = device, image
        self.record_every = record_every or (1000 if self.par["inter"] == "EFS" else 80)
        self.numTimeStep = self.par["steps"]
        self.records = 0

    def initializeDomainBorder(self):
        p = self.par
        if p["image"]:
            img = self._image
            if img is None:
                from .RKD2Q9 import load_structure_image
                img = load_structure_image(os.path.expanduser("~/StructureImage/structure.png"))
            self.isDomain = image_domain(img, 20, 0.5, duplicate=self.duplicate or p["duplicate"])
        else:
            self.isDomain = simple_geometry(p["nx"], p["ny"])
        self.ny, self.nx = self.isDomain.shape

    def initializeDomainCondition(self):
        p = self.par
        ii = np.mgrid[0:self.ny, 0:self.nx][0]
        lower = ii < (self.ny - 20 if p["image"] else self.ny - 10)
        fluid = self.isDomain == 1
        r = np.zeros((2, self.ny, self.nx))
        r[0][fluid & lower] = p["rho0"]; r[1][fluid & lower] = p["bg1"]
        r[1][fluid & ~lower] = p["rho1"]; r[0][fluid & ~lower] = p["bg0"]
        if p["cycle"]:
            # next drainage / imbibition cycle (ShanChenD2Q9.py:788-815): the old fluid keeps its last recorded
            # distribution below the top 30 rows, the new fluid fills those rows; populations at rest
            from .results import load_results
            old = None
            for ext in (".h5", ".npz"):
                f = os.path.join(self.initial_dir, "SimulationResults" + ext)
                if os.path.isfile(f):
                    old = load_results(f)
                    break
            key = "/FluidMacro/FluidDensityType0in%d" % p["last_step"]
            if old is None or key not in old:
                raise config.ConfigError("[DICycles] Option = 'yes': %s of SimulationResults not found in %s" % (key, self.initial_dir))
            prev = np.asarray(old[key], dtype=np.float64)
            if prev.shape != self.isDomain.shape:
                raise config.ConfigError("[DICycles]: recorded density has shape %s, the domain is %s" % (prev.shape, self.isDomain.shape))
            top = ii >= self.ny - 30
            r[0] = np.where(top, p["bg0"], prev)
            r[1] = np.where(top, p["rho1"], p["bg1"])
            r[:, ~fluid] = 0.0
        self.fluidsDensity = r

    def runTypeSCmodel(self, progress=None):
        p = self.par
        self.initializeDomainBorder()
        self.initializeDomainCondition()
        keys = ("inter", "relax", "tau0", "tau1", "G", "Gs0", "Gs1", "outlet", "method", "vy0", "vy1", "scheme")
        solver = SC2DSolver(self.isDomain, {k: p[k] for k in keys}, device=self.device, diagnostics=True)
        solver.set_density(self.fluidsDensity[0], self.fluidsDensity[1])
        out = ResultFile(self.output_dir, "SimulationResults",
                         (("FluidMacro", "MacroData"), ("FluidVelocity", "MacroVelocity")))
        self.result_path = out.path
        self._guard = RecordGuard("sc2d", int((self.isDomain == 1).sum()), getattr(self, "nan_guard", "raise"))
        total = self.numTimeStep + 1            # both reference loops run numTimeStep + 1 passes
        done = 0
        efs = p["inter"] == "EFS"
        while done < total:
            if efs:
                # EFS records at the END of loop pass i when i % 1000 == 0 (ShanChenD2Q9.py:2029)
                n = 1 if done == 0 else min(self.record_every, total - done)
                solver.step(n)
                done += n
                if (done - 1) % self.record_every == 0:
                    self._step_now = done
                    self._record(solver, out, ("rho0", "rho1"))
            else:
                # original Shan-Chen records inside pass `done+1`, after its inlet kernels, with the
                # velocity of the previous pass (ShanChenD2Q9.py:1523-1572)
                if done % self.record_every == 0:
                    self._step_now = done
                    self._record(solver, out, ("rec_rho0", "rec_rho1"))
                n = min(self.record_every - done % self.record_every, total - done)
                solver.step(n)
                done += n
            if progress:
                progress(done)
        solver.sync()
        self.solver = solver
        return self.result_path

    def _record(self, solver, out, rho_fields):
        k = self.records
        self.fluidsDensity = np.stack([solver.get(rho_fields[0]), solver.get(rho_fields[1])])
        self.physicalVX, self.physicalVY = solver.get("vx"), solver.get("vy")
        for i in range(2):
            out.write("FluidMacro", "FluidDensityType%gin%g" % (i, k), self.fluidsDensity[i])
        out.write("FluidVelocity", "FluidVelocityXAt%g" % k, self.physicalVX)
        out.write("FluidVelocity", "FluidVelocityYAt%g" % k, self.physicalVY)
        guard = getattr(self, "_guard", None)
        if guard:
            guard(k, getattr(self, "_step_now", 0), dict(rho0=self.fluidsDensity[0], rho1=self.fluidsDensity[1], vx=self.physicalVX, vy=self.physicalVY),
                  dict(mass0=float(self.fluidsDensity[0].sum()), mass1=float(self.fluidsDensity[1].sum())))
        self.records += 1
