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
from .results import RecordGuard, ResultFile
from .rk2d import RK2DSolver


def load_structure_image(path):
    """Greyscale float array of a pore image, 0 = solid (replaces scipy.ndimage.imread(path, True),
    removed from SciPy; RKD2Q9.py:382)."""
    from PIL import Image
    return np.asarray(Image.open(path).convert("L"), dtype=np.float64)


class RKColorGradientLBM:
    def __init__(self, pathIniFile, output_dir=None, image=None, device=0, initial_dir=None):
        self.pathIni = pathIniFile
        self.initial_dir = initial_dir or os.path.expanduser("~/LBMInitial")
        self.fluidPDFR = self.fluidPDFB = None
        self.physicalVX = self.physicalVY = None
        self.par = config.read_rk2d(pathIniFile)
        self.output_dir = output_dir or os.path.expanduser("~/LBMResults")
        self.device = device
        self.nan_guard = "raise"              # 'raise' | 'warn' | 'off': finiteness check of every record (results.RecordGuard)
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
            self._initial_state_from_files()
            return
        self.fluidsRhoR, self.fluidsRhoB = initial_densities_rk(self.isDomain, p["image"], p["nbuf"],
                                                                 p["rho0R"], p["rho0B"])

    def _initial_state_from_files(self):
        """[CyclesSetup] IsCycle = 'yes' (drainage-imbibition cycles, RKD2Q9.py:491-559): start from the
        results of a previous run kept in ~/LBMInitial."""
        from .results import load_results
        p = self.par

        def find(stem):
            for ext in (".h5", ".npz"):
                f = os.path.join(self.initial_dir, stem + ext)
                if os.path.isfile(f):
                    return load_results(f)
            raise config.ConfigError("IsCycle = 'yes': no %s.h5/.npz in %s" % (stem, self.initial_dir))

        def need(d, key):
            if key not in d:
                raise config.ConfigError("IsCycle = 'yes': dataset %s missing" % key)
            a = np.array(d[key], dtype=np.float64)
            if a.shape[:2] != self.isDomain.shape:
                raise config.ConfigError("IsCycle = 'yes': %s has shape %s, the domain is %s" % (key, a.shape, self.isDomain.shape))
            return a

        if not p["image"]:
            # last record of SimulationResultsRK; top 20 rows refilled with blue; f = f_eq(rho, u) (:492-508)
            d, k = find("SimulationResultsRK"), p["last_step"]
            self.fluidsRhoR = need(d, "/FluidMacro/FluidDensityRin%d" % k)
            self.fluidsRhoB = need(d, "/FluidMacro/FluidDensityBin%d" % k)
            self.physicalVX = need(d, "/FluidVelocity/FluidVelocityXAt%d" % k)
            self.physicalVY = need(d, "/FluidVelocity/FluidVelocityYAt%d" % k)
            self.fluidsRhoR[-20:, :] = 0.0
            self.fluidsRhoB[-20:, :] = p["rho0B"]
            solid = self.isDomain != 1
            for a in (self.fluidsRhoR, self.fluidsRhoB, self.physicalVX, self.physicalVY):
                a[solid] = 0.0
        else:
            # cycleInitialRK: populations taken over, colours swapped in the top buffer rows (:532-556)
            d, nb = find("cycleInitialRK"), p["nbuf"]
            rR, rB = need(d, "/FluidMacro/FluidDensityR"), need(d, "/FluidMacro/FluidDensityB")
            fR, fB = need(d, "/FluidPDF/FluidPDFR"), need(d, "/FluidPDF/FluidPDFB")
            self.fluidsRhoR, self.fluidsRhoB = rR.copy(), rB.copy()
            self.fluidPDFR, self.fluidPDFB = fR.copy(), fB.copy()
            self.fluidsRhoR[-nb:], self.fluidsRhoB[-nb:] = rB[-nb:], rR[-nb:]
            self.fluidPDFR[-nb:], self.fluidPDFB[-nb:] = fB[-nb:], fR[-nb:]
            self.physicalVX = need(d, "/FluidVelocity/FluidVelocityX")
            self.physicalVY = need(d, "/FluidVelocity/FluidVelocityY")

    def _upload_initial_state(self, solver):
        if self.fluidPDFR is not None:
            solver.set_pdf(self.fluidPDFR, self.fluidPDFB)
        else:
            solver.set_macro(self.fluidsRhoR, self.fluidsRhoB, self.physicalVX, self.physicalVY)

    # -- run
    def runRKColorGradient2D(self, progress=None):
        p = self.par
        self.initializeDomainBorder()
        self.initializeDomainCondition()
        keys = ("sigma", "theta", "wetting", "beta", "delta", "tauR", "tauB", "tautype", "relax", "inlet", "outlet",
                "vyR", "vyB", "rhoBH", "rhoRH", "rhoBL", "rhoRL")
        solver = RK2DSolver(self.isDomain, {k: p[k] for k in keys}, device=self.device)
        self._upload_initial_state(solver)
        out = ResultFile(self.output_dir, "SimulationResultsRK",
                         (("FluidMacro", "MacroData"), ("FluidPDF", "MicroData"), ("FluidVelocity", "MacroVelocity")))
        self.result_path = out.path
        self._guard = RecordGuard("rk2d", int((self.isDomain == 1).sum()), self.nan_guard)
        done = 0
        while done < self.timeSteps:
            self._step_now = done
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
        guard = getattr(self, "_guard", None)
        if guard:
            total = float(self.fluidsRhoR.sum() + self.fluidsRhoB.sum())
            guard(k, getattr(self, "_step_now", 0), dict(rhoR=self.fluidsRhoR, rhoB=self.fluidsRhoB, vx=self.physicalVX, vy=self.physicalVY),
                  dict(massR=float(self.fluidsRhoR.sum()), massB=float(self.fluidsRhoB.sum()),
                       saturationR=float(self.fluidsRhoR.sum()) / total if total else 0.0))
        self.records += 1
