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
        if p["tension_type"] == "Perturbation":
            return self.runRKColorGradient2DPerturbation(progress)
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

    # -- the perturbation loop (RKD2Q9.py:978-1223), kernel by kernel on the kernel-level layer
    def runRKColorGradient2DPerturbation(self, progress=None, initial_pdf=None):
        """[SurfaceTension] SurfaceTensionType = 'Perturbation'.  The reference's method of this name stops at its first
        inlet launch (RKD2Q9.py:1099: 10 arguments for a 12-argument kernel); this is that loop with the four call-site
        repairs under which its golden captures were taken (tests/golden/gen/make_golden_rk_pert.py):
          R1 the two force arrays ghostPointsConstantVelocityRK also takes are passed (unused by the kernel);
          R2 calPhysicalVelocityRKGPU2D gets its eight arguments (the call at :1120 adds two);
          R3 f_tot = f_R + f_B is taken where the collision kernels need it (after collision 1 for SRT, before it for MRT);
          R4 body force zero.
        Also left out: the loop's last launch, calRecoloringProcess (:1206), reads gradient and collision arrays nothing ever
        writes -- it adds zeros where device_array_like happens to return zeros, garbage otherwise.
        Schedule (`self.perturbation_schedule`): "fused" = one launch per time step (rk2dp_fused behind lbmpm_rk2d_set_perturbation:
        every boundary type of the loop, no solid node in the boundary rows); "kernels" = the loop kernel by kernel, ~16
        launches per step on the kernel-level entry points (include/lbmpm_kernels.h), arrays in the reference's sparse layout;
        "auto" (default) = fused where it applies, else kernels.
        `initial_pdf` = (fR, fB) dense [ny][nx][9] replaces the rest-state start (tests)."""
        import sys
        drop = os.path.join(os.path.dirname(os.path.abspath(__file__)), "dropin")
        if drop not in sys.path:
            sys.path.append(drop)
        import _runtime as rt
        p = self.par
        if p["tension_type"] != "Perturbation":
            raise config.ConfigError("runRKColorGradient2DPerturbation needs SurfaceTensionType = 'Perturbation'")
        self.initializeDomainBorder()
        self.initializeDomainCondition()
        schedule = getattr(self, "perturbation_schedule", "auto")
        if schedule not in ("auto", "fused", "kernels"):
            raise ValueError("perturbation_schedule must be 'auto', 'fused' or 'kernels'")
        # `self.perturbation_order = "literal"` (opt-in; default "repaired"): the loop WITHOUT repair R3 -- calTotalFluidPDF where
        # RKD2Q9.py:1065 has it, right after streaming, so that calRKCollision23GPUNew recolours from the pre-boundary, pre-relaxation
        # sum.  Kernel by kernel only; tests/golden/rkpert_*_literal.npz are captures of the reference in that order.
        order = getattr(self, "perturbation_order", "repaired")
        if order not in ("repaired", "literal"):
            raise ValueError("perturbation_order must be 'repaired' or 'literal'")
        if order == "literal":
            if schedule == "fused":
                raise ValueError("the fused perturbation step implements the repaired order; the literal order runs kernel by kernel")
            schedule = "kernels"
        if schedule != "kernels":
            done = self._run_perturbation_fused(progress, initial_pdf, required=schedule == "fused")
            if done:
                return self.result_path
        ny, nx = self.isDomain.shape
        fluid = np.flatnonzero(self.isDomain.reshape(-1) == 1).astype(np.int64)          # optimizeFluidArray, :603-655
        N = int(fluid.size)
        new_index = -np.ones(nx * ny, dtype=np.int64); new_index[fluid] = np.arange(N)
        W = np.array([4. / 9.] + [1. / 9.] * 4 + [1. / 36.] * 4)
        ex = np.array([0., 1., 0., -1., 0., 1., -1., -1., 1.]); ey = np.array([0., 0., 1., 0., -1., 1., 1., -1., -1.])
        rR = self.fluidsRhoR.reshape(-1)[fluid]; rB = self.fluidsRhoB.reshape(-1)[fluid]
        if initial_pdf is not None:
            fR = np.ascontiguousarray(np.asarray(initial_pdf[0], dtype=np.float64).reshape(-1, 9)[fluid])
            fB = np.ascontiguousarray(np.asarray(initial_pdf[1], dtype=np.float64).reshape(-1, 9)[fluid])
        else:
            fR = np.ascontiguousarray(rR[:, None] * W[None, :]); fB = np.ascontiguousarray(rB[:, None] * W[None, :])   # :577-601, u = 0
        dev = lambda a: rt.to_device(np.ascontiguousarray(a))
        z = lambda *shape: dev(np.zeros(shape))
        mrt = p["relax"] == "MRT"
        T = dict(totalNodes=N, totalNum=N, nx=nx, ny=ny, xDim=128, fluidNodes=dev(fluid), domainNewIndex=dev(new_index),
                 neighboringNodes=dev(np.zeros(8 * N, dtype=np.int64)), fluidPDFR=dev(fR), fluidPDFB=dev(fB), fluidPDFRNew=z(N, 9),
                 fluidPDFBNew=z(N, 9), fluidPDFTotal=dev(fR + fB), fluidRhoR=dev(rR), fluidRhoB=dev(rB), physicalVX=z(N), physicalVY=z(N),
                 phiValue=z(N), forceX=z(N), forceY=z(N), collisionR1=z(N, 9), collisionB1=z(N, 9), CGX=z(nx * ny), CGY=z(nx * ny),
                 unitEX=dev(ex), unitEY=dev(ey), weightsCoeff=dev(W), schemeGradient=dev(np.ones(9)),
                 constantCR=z(9), constantCB=z(9),                         # rest weights C_i(alpha): loaded and never used by these kernels
                 constantB=dev(np.array([-2. / 9.] + [1. / 9.] * 4 + [1. / 36.] * 4)),              # constantBNew, :131-133
                 delta=p["delta"], tauR=p["tauR"], tauB=p["tauB"], betaCoeff=p["beta"], AkR=p["AkR"], AkB=p["AkB"], solidPhi=p["solidPhi"],
                 specificVYR=p["vyR"], specificVYB=p["vyB"], constPLB=p["rhoBL"], constPLR=p["rhoRL"], constPHB=p["rhoBH"], constPHR=p["rhoRH"],
                 bodyFX=0.0, bodyFY=0.0)
        if mrt:
            from .rk2d import mrt_matrices
            M, Minv, S = mrt_matrices()
            T.update(transformationM=dev(M), inverseTM=dev(Minv), collisionS=dev(S))
        go = lambda name, **rename: rt.launch_by_name("rk", name, dict(T, **{k: T[v] for k, v in rename.items()}))
        go("fillNeighboringNodes")                                       # non-fluid neighbours come out as -1 (new_index)
        R = dict(fluidPDF="fluidPDFR", fluidPDFNew="fluidPDFRNew"); Bq = dict(fluidPDF="fluidPDFB", fluidPDFNew="fluidPDFBNew")
        outlet = ([("convectiveOutletGPU", {}), ("convectiveOutletGhost2GPU", {}), ("convectiveOutletGhost3GPU", {})] if p["outlet"] == "Convective"
                  else [("calConstPressureLowerGPU", {}), ("ghostPointsConstPressureLowerRK", {})])                      # :1064-1088
        inlet = ([("constantVelocityZHBoundaryHigherRK", {}), ("ghostPointsConstantVelocityRK", {})] if p["inlet"] == "Neumann"
                 else [("calConstPressureInletGPU", {}), ("ghostPointsConstPressureInletRK", {})])                       # :1089-1110
        literal = [("calTotalFluidPDF", {})] if order == "literal" else []       # RKD2Q9.py:1065
        head = [("calStreaming1GPU", R), ("calStreaming1GPU", Bq), ("calStreaming2GPU", R), ("calStreaming2GPU", Bq)] + literal + outlet + inlet + \
               [("calMacroDensityRKGPU2D", {}), ("calPhysicalVelocityRKGPU2D", {})]
        if order == "literal":
            tail = [("calPhaseFieldPhi", {}), ("calRKCollision1GPU2DMRTNew" if mrt else "calRKCollision1GPU2DSRTNew", {}), ("calRKCollision23GPUNew", {})]
        else:
            tail = [("calPhaseFieldPhi", {})] + ([("calTotalFluidPDF", {}), ("calRKCollision1GPU2DMRTNew", {})] if mrt
                                                 else [("calRKCollision1GPU2DSRTNew", {}), ("calTotalFluidPDF", {})]) + [("calRKCollision23GPUNew", {})]
        out = ResultFile(self.output_dir, "SimulationResultsRK",
                         (("FluidMacro", "MacroData"), ("FluidPDF", "MicroData"), ("FluidVelocity", "MacroVelocity")))
        self.result_path = out.path
        self._guard = RecordGuard("rk2d perturbation", N, self.nan_guard)

        def dense(compact, tail_shape=()):
            a = np.zeros((nx * ny,) + tail_shape)
            a[fluid] = compact
            return a.reshape((ny, nx) + tail_shape)

        class _View:                                                     # what _record reads from a fused solver
            def get(_, name):
                src = dict(rec_rhoR="fluidRhoR", rec_rhoB="fluidRhoB", rec_vx="physicalVX", rec_vy="physicalVY", rec_fR="fluidPDFR", rec_fB="fluidPDFB")[name]
                a = T[src].copy_to_host()
                return dense(a, (9,) if a.ndim == 2 else ())
        view = _View()
        self._pert_table, self.fluidNodes = T, fluid
        for step in range(1, self.timeSteps + 1):
            self._step_now = step - 1
            for name, rename in head:
                go(name, **rename)
            if (step - 1) % self.timeInterval == 0:                      # :1121-1131
                self._record(view, out)
            for name, rename in tail:
                go(name, **rename)
            if progress:
                progress(step)
        return self.result_path

    def _run_perturbation_fused(self, progress, initial_pdf, required):
        """the perturbation loop on the fused solver; False when the fused step does not cover this set-up (and it was not demanded)"""
        from ._lib import LbmpmError
        from .rk2d import RK2DSolver
        p = self.par
        par = dict(beta=p["beta"], tauR=p["tauR"], tauB=p["tauB"], relax=p["relax"], inlet=p["inlet"], outlet=p["outlet"], vyR=p["vyR"], vyB=p["vyB"],
                   rhoRL=p["rhoRL"], rhoBL=p["rhoBL"], rhoRH=p["rhoRH"], rhoBH=p["rhoBH"])
        try:
            solver = RK2DSolver(self.isDomain, par, perturbation=dict(AkR=p["AkR"], AkB=p["AkB"], solidPhi=p["solidPhi"]))
        except LbmpmError as e:
            from ._lib import ERR_UNSUPPORTED
            if required or e.status != ERR_UNSUPPORTED:                  # LBMPM_ERR_UNSUPPORTED: the kernel-level loop covers it
                raise
            return False
        ny, nx = self.isDomain.shape
        fluid = self.isDomain == 1
        W = np.array([4. / 9.] + [1. / 9.] * 4 + [1. / 36.] * 4)
        if initial_pdf is not None:
            fR, fB = (np.asarray(a, dtype=np.float64).reshape(ny, nx, 9) for a in initial_pdf)
        else:                                                             # :577-601 with u = 0
            fR = np.where(fluid[..., None], self.fluidsRhoR[..., None] * W, 0.0); fB = np.where(fluid[..., None], self.fluidsRhoB[..., None] * W, 0.0)
        solver.set_pdf(fR, fB)
        out = ResultFile(self.output_dir, "SimulationResultsRK",
                         (("FluidMacro", "MacroData"), ("FluidPDF", "MicroData"), ("FluidVelocity", "MacroVelocity")))
        self.result_path = out.path
        self._guard = RecordGuard("rk2d perturbation", int(fluid.sum()), self.nan_guard)
        self.fluidNodes = np.flatnonzero(self.isDomain.reshape(-1) == 1).astype(np.int64)
        step = 1
        while step <= self.timeSteps:
            self._step_now = step - 1
            if (step - 1) % self.timeInterval == 0:                      # :1121-1131: after the step's streaming, boundary kernels, velocity
                self._record(solver, out)
            nxt = min(self.timeSteps + 1, ((step - 1) // self.timeInterval + 1) * self.timeInterval + 1)
            solver.step(nxt - step)
            step = nxt
            if progress:
                progress(step - 1)
        solver.sync()
        self.solver = solver
        return True

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
