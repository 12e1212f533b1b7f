"""Driver counterpart of the reference's RKCG2D/RKD2Q9.py (own code, same ini, same outputs):
class RKColorGradientLBM(pathIniFile).runRKColorGradient2D().

What is kept: ini grammar (openlbmpm_amd/config.py), geometry rules (SimpleGeometry walls or the
image crop/wall/buffer rules), initial fields (RKD2Q9.py:459-531), record cadence
`(iStep-1) % TimeInterval == 0` with the record counter as dataset suffix and the dataset names
of resultInHDF5 (RKD2Q9.py:938-957).  What is dropped: prints before every launch, input()
pauses, PNG plots, the sparse compaction (implicit in the library's dense mask).
The time loop itself is `lbmpm_rk2d_step` (liblbmpm_hip.so).
"""
import os

import numpy as np

from . import config
from .geometry import simple_geometry, image_domain, initial_densities_rk
from .results import ResultFile
from .rk2d import RK2DSolver


def load_structure_image(path):
    """Greyscale float array of a pore image, 0 = solid (replaces scipy.ndimage.imread(path, True),
    removed from SciPy; RKD2Q9.py:382)."""
    from PIL import Image
    return np.asarray(Image.open(path).convert("L"), dtype=np.float64)


class RKColorGradientLBM:
    def __init__(self, pathIniFile, output_dir=None, image=None, device=0):
        self.pathIni = pathIniFile
        self.par = config.read_rk2d(pathIniFile)
        self.output_dir = output_dir or os.path.expanduser("~/LBMResults")
        self.device = device
        self._image = image
        p = self.par
        self.timeSteps, self.timeInterval = p["steps"], p["interval"]
        self.isDomain = None
        self.records = 0

    # -- set-up (RKD2Q9.py:417-443, :445-601)
    def initializeDomainBorder(self):
        p = self.par
        if p["image"]:
            img = self._image
            if img is None:
                img = load_structure_image(os.path.expanduser("~/StructureImage/structure.png"))
            self.isDomain = image_domain(img, p["nbuf"], p["ratio"])
        else:
            self.isDomain = simple_geometry(p["nx"], p["ny"])
        self.yDomain, self.xDomain = self.isDomain.shape
        self.voidSpace = int(np.count_nonzero(self.isDomain))

    def initializeDomainCondition(self):
        p = self.par
        if p["cycle"]:
            raise config.ConfigError("IsCycle = 'yes' (restart from ~/LBMInitial) is not wired up yet")
        self.fluidsRhoR, self.fluidsRhoB = initial_densities_rk(self.isDomain, p["image"], p["nbuf"],
                                                                 p["rho0R"], p["rho0B"])

    # -- run
    def runRKColorGradient2D(self, progress=None):
        p = self.par
        self.initializeDomainBorder()
        self.initializeDomainCondition()
        keys = ("sigma", "theta", "wetting", "beta", "delta", "tauR", "tauB", "tautype", "relax", "inlet", "outlet",
                "vyR", "vyB", "rhoBH", "rhoRH", "rhoBL", "rhoRL")
        solver = RK2DSolver(self.isDomain, {k: p[k] for k in keys}, device=self.device)
        solver.set_macro(self.fluidsRhoR, self.fluidsRhoB)
        out = ResultFile(self.output_dir, "SimulationResultsRK",
                         (("FluidMacro", "MacroData"), ("FluidPDF", "MicroData"), ("FluidVelocity", "MacroVelocity")))
        self.result_path = out.path
        done = 0
        while done < self.timeSteps:
            # the reference records inside step `done+1`, after its boundary kernels, velocity and
            # phase field (RKD2Q9.py:1382-1393): that is the REC_* view of the current state
            if done % self.timeInterval == 0:
                self._record(solver, out)
            n = min(self.timeInterval - done % self.timeInterval, self.timeSteps - done)
            solver.step(n)
            done += n
            if progress:
                progress(done)
        solver.sync()
        self.solver = solver
        return self.result_path

    def _record(self, solver, out):
        k = self.records
        self.fluidsRhoR = solver.get("rec_rhoR"); self.fluidsRhoB = solver.get("rec_rhoB")
        self.physicalVX = solver.get("rec_vx"); self.physicalVY = solver.get("rec_vy")
        self.fluidPDFR = solver.get("rec_fR"); self.fluidPDFB = solver.get("rec_fB")
        out.write("FluidMacro", "FluidDensityRin%g" % k, self.fluidsRhoR)
        out.write("FluidMacro", "FluidDensityBin%g" % k, self.fluidsRhoB)
        out.write("FluidPDF", "FluidPDFBat%g" % k, self.fluidPDFB)
        out.write("FluidPDF", "FluidPDFRat%g" % k, self.fluidPDFR)
        out.write("FluidVelocity", "FluidVelocityXAt%g" % k, self.physicalVX)
        out.write("FluidVelocity", "FluidVelocityYAt%g" % k, self.physicalVY)
        self.records += 1
