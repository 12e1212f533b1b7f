"""3-D colour-gradient driver: class RKColorGradient3D(pathIniFile).runRKColorGradient3D() -- the
entry the reference's main.py:22,80-81 calls but whose module is missing from its tree.  Own
design by analogy with the 2-D driver (RKD2Q9.py): reads IniFiles/RKtwophasesetup3D.ini, builds a
duct (solid side walls, open z planes) or takes a voxel array, starts red below the top buffer
planes and blue in them (RKD2Q9.py:511-531 carried to 3-D), records densities and velocity.

One process per GPU: when torch.distributed is initialised with world size > 1 the lattice is cut
into z-slabs (openlbmpm_amd/rk3d.py: RK3DDistributed, halos over RCCL) and every rank writes the
planes it owns to its own file; otherwise a single slab on `device`.
"""
import os

import numpy as np

from . import config
from .geometry import initial_densities_rk3d, voxel_domain
from .results import RecordGuard, ResultFile
from .rk3d import RK3DSlab, RK3DDistributed

PARAM_KEYS = ("AkR", "AkB", "beta", "tauR", "tauB", "SolidRhoR", "SolidRhoB", "velocityZR", "velocityZB",
              "densityRL", "densityBL", "relax")


def duct(nx, ny, nz):
    """[nz][ny][nx] mask: solid walls on the four sides, open inlet / outlet planes in z"""
    dom = np.ones((nz, ny, nx), dtype=np.uint8)
    dom[:, 0, :] = dom[:, -1, :] = 0
    dom[:, :, 0] = dom[:, :, -1] = 0
    return dom


class RKColorGradient3D:
    def __init__(self, pathIniFile, output_dir=None, domain=None, device=0, record_every=None, num_buffering_layers=10,
                 structure_path=None):
        self.pathIni = pathIniFile
        self.par = config.read_rk3d(pathIniFile)
        self.output_dir = output_dir or os.path.expanduser("~/LBMResults3D")       # main.py:28
        self.device, self._domain, self.nbuf = device, domain, int(num_buffering_layers)
        self.structure_path = structure_path
        self.timeSteps = self.par["steps"]
        self.timeInterval = record_every or self.par["interval"] or max(1, self.timeSteps // 10)
        self.records = 0

    def initializeDomainBorder(self):
        p = self.par
        if self._domain is not None:
            self.isDomain = np.ascontiguousarray(self._domain, dtype=np.uint8)
            if self.isDomain.ndim != 3:
                raise TypeError("domain must be a [nz][ny][nx] array")
        elif p["image"]:
            # the reference ships no 3-D image reader; a NumPy voxel file next to where its 2-D drivers look
            # for structure.png (non-zero = pore), framed like the 2-D images are
            path = self.structure_path or os.path.expanduser("~/StructureImage/structure3D.npy")
            if not os.path.isfile(path):
                raise config.ConfigError("[ImageSetup] Existance = 'yes': voxel file %s not found (or pass `domain=`)" % path)
            self.isDomain = voxel_domain(np.load(path), self.nbuf)
        else:
            self.isDomain = duct(p["nx"], p["ny"], p["nz"])
        self.zDomain, self.yDomain, self.xDomain = self.isDomain.shape
        self.voidSpace = int(np.count_nonzero(self.isDomain))

    def initializeDomainCondition(self):
        p = self.par
        self.fluidsRhoR, self.fluidsRhoB = initial_densities_rk3d(self.isDomain, min(self.nbuf, self.zDomain // 4),
                                                                  p["rho0R"], p["rho0B"])

    def _distributed(self):
        try:
            import torch.distributed as dist
            return dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1
        except ImportError:
            return False

    def runRKColorGradient3D(self, progress=None):
        p = self.par
        self.initializeDomainBorder()
        self.initializeDomainCondition()
        par = {k: p[k] for k in PARAM_KEYS}
        name = "SimulationResultsRK3D"
        if self._distributed():
            import torch.distributed as dist
            sim = RK3DDistributed(self.isDomain, par, device=self.device)
            sim.set_density(self.fluidsRhoR, self.fluidsRhoB)
            if getattr(self, "calibrate_partition", self.timeSteps >= 1000):
                # long runs: one measured re-cut of the slabs (rk3d.RK3DDistributed.calibrated_plane_cost), then start over from the
                # initial state -- the rank that holds the colour interface otherwise runs ~10 % behind the others
                cost = sim.calibrated_plane_cost(8)
                sim.close()
                sim = RK3DDistributed(self.isDomain, par, device=self.device, plane_cost=cost)
                sim.set_density(self.fluidsRhoR, self.fluidsRhoB)
            step, observe, slab = sim.step, sim.observe, sim.slab
            name += "_rank%d" % dist.get_rank()
            self.z0, self.nzl = sim.z0, sim.nzl
        else:
            slab = sim = RK3DSlab(self.isDomain, 0, self.zDomain, par, self.device)
            slab.set_density(self.fluidsRhoR, self.fluidsRhoB)
            step = slab.step_single
            observe = lambda: slab.phase_field(diagnostics=True)
            self.z0, self.nzl = 0, self.zDomain
        out = ResultFile(self.output_dir, name, (("FluidMacro", "MacroData"), ("FluidVelocity", "MacroVelocity")))
        self.result_path = out.path
        # distributed: every rank checks its own slab, the verdict is collective (all ranks raise together, none is left in an exchange)
        self._guard = RecordGuard("rk3d", slab.num_fluid_nodes, getattr(self, "nan_guard", "raise"), collective=self._distributed(), device=self.device)
        done = 0
        while done < self.timeSteps:
            self._step_now = done
            if done % self.timeInterval == 0:
                observe()
                self._record(slab, out)
            n = min(self.timeInterval - done % self.timeInterval, self.timeSteps - done)
            step(n)
            done += n
            if progress:
                progress(done)
        observe()
        self._step_now = done
        self._record(slab, out)
        slab.sync()
        self.solver = sim
        return self.result_path

    def _record(self, slab, out):
        k = self.records
        self.fluidsRhoR, self.fluidsRhoB = slab.get("rhoR"), slab.get("rhoB")
        self.physicalVX, self.physicalVY, self.physicalVZ = slab.get("vx"), slab.get("vy"), slab.get("vz")
        out.write("FluidMacro", "FluidDensityRin%g" % k, self.fluidsRhoR)
        out.write("FluidMacro", "FluidDensityBin%g" % k, self.fluidsRhoB)
        for axis, a in (("X", self.physicalVX), ("Y", self.physicalVY), ("Z", self.physicalVZ)):
            out.write("FluidVelocity", "FluidVelocity%sAt%g" % (axis, k), a)
        guard = getattr(self, "_guard", None)
        if guard:
            guard(k, getattr(self, "_step_now", 0), dict(rhoR=self.fluidsRhoR, rhoB=self.fluidsRhoB, vx=self.physicalVX, vy=self.physicalVY, vz=self.physicalVZ),
                  dict(massR=float(self.fluidsRhoR.sum()), massB=float(self.fluidsRhoB.sum())))
        self.records += 1
