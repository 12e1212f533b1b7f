"""3-D colour-gradient driver: class RKColorGradient3D(pathIniFile).runRKColorGradient3D() -- the
entry the reference's main.py:22,80-81 calls but whose module is missing from its tree.  Own
design by analogy with the 2-D driver (RKD2Q9.py): reads IniFiles/RKtwophasesetup3D.ini, builds a
duct (solid side walls, open z planes) or takes a voxel array, starts red below the top buffer
planes and blue in them (RKD2Q9.py:511-531 carried to 3-D), records densities and velocity
(RKD2Q9.py:938-957; the populations too with record_pdf=True).

[CyclesSetup] IsCycle = 'yes' (RKtwophasesetup3D.ini:57-59) follows the 2-D rules (RKD2Q9.py:491-559) with z in the place of y:
  * no image: the record LastStep of <initial_dir>/SimulationResultsRK3D (densities + velocity), the top 20 planes refilled with
    blue, populations = equilibria of those fields (RKD2Q9.py:492-508),
  * image: <initial_dir>/cycleInitialRK3D -- /FluidMacro/FluidDensityR|B, /FluidPDF/FluidPDFR|B [nz][ny][nx][19],
    /FluidVelocity/FluidVelocityX|Y|Z -- taken over with the colours swapped in the top buffer planes (RKD2Q9.py:532-556);
    write_cycle_initial() writes that file from a finished run.
Beyond the reference: checkpoint() / restart_from= keep the solver's stored state (lbmpm_rk3d_get_state / set_state): a run
continued from a checkpoint equals the uninterrupted one bit for bit, on any number of ranks.

[SurfaceTension] SurfaceTensionType = 'CSF' (the section of RKtwophasesetup2D.ini added to the 3-D file): the 2-D CSF loop carried to
D3Q19 (openlbmpm_amd/rk3dcsf.py, lbmpm_rk3dcsf_*) instead of the perturbation loop -- surface tension, contact angle (wetting rule 2),
DeltaValue, TauType from the ini; one GPU or one z-slab per rank (rk3dcsf.RK3DCSFDistributed: three face messages per step); records hold what the reference records (the lattice after the next step's boundary
planes, RKD2Q9.py:1382-1393); IsCycle and checkpoints as above (a checkpoint keeps the streamed populations and the last force).

One process per GPU: when torch.distributed is initialised with world size > 1 the lattice is cut
into z-slabs (openlbmpm_amd/rk3d.py: RK3DDistributed, halos over xGMI) and rank 0 writes ONE result file with the
whole lattice's arrays, gathered at the record cadence (gather_records = False: every rank its own planes in its own file);
otherwise a single slab on `device`.
"""
import os

import numpy as np

from . import config
from .geometry import initial_densities_rk3d, voxel_domain
from .results import RecordGuard, ResultFile, find_result_file, read_planes
from .rk3d import RK3DSlab, RK3DDistributed

PARAM_KEYS = ("AkR", "AkB", "beta", "tauR", "tauB", "SolidRhoR", "SolidRhoB", "velocityZR", "velocityZB",
              "densityRL", "densityBL", "relax", "inlet", "densityRH", "densityBH", "outlet")
GROUPS = (("FluidMacro", "MacroData"), ("FluidPDF", "MicroData"), ("FluidVelocity", "MacroVelocity"))


class _CSFSlab:
    """RK3DCSFSolver behind the calls this driver makes on a slab of the perturbation model"""

    def __init__(self, dom, par, device, bulk_epsilon=0.0, distributed=False):
        from .rk3dcsf import RK3DCSFSolver, RK3DCSFDistributed
        q = dict(sigma=par["sigma"], theta=par["theta"], wetting=par["wetting"], beta=par["beta"], delta=par["delta"], tauR=par["tauR"], tauB=par["tauB"],
                 tautype=par["tautype"], relax=par["relax"], inlet=par["inlet"], outlet=par["outlet"], velocityZR=par["velocityZR"],
                 velocityZB=par["velocityZB"], densityBH=par["densityBH"], densityRH=par["densityRH"], densityBL=par["densityBL"], densityRL=par["densityRL"],
                 bulk_epsilon=float(bulk_epsilon))
        # distributed: one slab per rank; set_* take the undivided arrays (a slab cuts its planes and the images of its neighbours' out of
        # them), get* return the rank's own planes
        self.solver = RK3DCSFDistributed(dom, q, device=device) if distributed else RK3DCSFSolver(dom, q, device=device)
        self.step_single, self.sync, self.close = self.solver.step, self.solver.sync, self.solver.close

    num_fluid_nodes = property(lambda self: self.solver.num_fluid_nodes)
    dominant_kernel = property(lambda self: self.solver.dominant_kernel)

    def set_density(self, rR, rB):
        self.solver.set_macro(rR, rB)

    def set_macro(self, rR, rB, vx, vy, vz):
        self.solver.set_macro(rR, rB, vx, vy, vz)

    def set_pdf(self, fR, fB, post_collision=True):
        self.solver.set_pdf(fR, fB)          # the recorded arrays are the loop's arrays at its top: streamed populations

    def get(self, name):
        return self.solver.get("rec_" + name)

    def get_pdf(self):
        return self.solver.get("rec_fR"), self.solver.get("rec_fB")

    def get_state(self):
        s = self.solver
        st = np.concatenate([s.get("fR"), s.get("fB")] + [s.get(c)[..., None] for c in ("Fx", "Fy", "Fz")], axis=-1)
        return st, dict(doubles_per_cell=41, steps=s.steps_done, post_collision=False)

    def set_state(self, st, steps, post_collision):
        st = np.asarray(st)
        if st.shape[-1] != 41:
            raise config.ConfigError("restart_from: this checkpoint is not one of the 3-D CSF model (41 doubles per cell: f_R, f_B, F)")
        self.solver.set_pdf(st[..., :19], st[..., 19:38], force=tuple(np.ascontiguousarray(st[..., 38 + a]) for a in range(3)))


def duct(nx, ny, nz):
    """[nz][ny][nx] mask: solid walls on the four sides, open inlet / outlet planes in z"""
    dom = np.ones((nz, ny, nx), dtype=np.uint8)
    dom[:, 0, :] = dom[:, -1, :] = 0
    dom[:, :, 0] = dom[:, :, -1] = 0
    return dom


class RKColorGradient3D:
    def __init__(self, pathIniFile, output_dir=None, domain=None, device=0, record_every=None, num_buffering_layers=10,
                 structure_path=None, initial_dir=None, record_pdf=False, restart_from=None, checkpoint_every=0, csf_bulk_epsilon=0.0):
        self.pathIni = pathIniFile
        self.par = config.read_rk3d(pathIniFile)
        self.output_dir = output_dir or os.path.expanduser("~/LBMResults3D")       # main.py:28
        self.initial_dir = initial_dir or os.path.expanduser("~/LBMInitial")       # RKD2Q9.py:492
        self.device, self._domain, self.nbuf = device, domain, int(num_buffering_layers)
        self.structure_path = structure_path
        self.timeSteps = self.par["steps"]
        self.timeInterval = record_every or self.par["interval"] or max(1, self.timeSteps // 10)
        self.record_pdf = bool(record_pdf)          # /FluidPDF/FluidPDFRat<k>, ...Bat<k> [nz][ny][nx][19] with every record (38 doubles per cell)
        self.restart_from, self.checkpoint_every = restart_from, int(checkpoint_every)
        self.csf_bulk_epsilon = float(csf_bulk_epsilon)      # 3-D CSF only, opt-in (include/lbmpm.h: lbmpm_rk3dcsf_config.bulk_epsilon; 0 = exact)
        self.gather_records = True
        self.records = 0
        self.physicalVX = self.physicalVY = self.physicalVZ = None
        self.fluidPDFR = self.fluidPDFB = None

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

    @property
    def _is_image(self):
        return self._domain is not None or self.par["image"]

    def initializeDomainCondition(self, z0=0, nzl=None):
        """initial fields of the planes [z0, z0 + nzl) (a rank's slab; default: the whole lattice)"""
        p = self.par
        nzl = self.zDomain if nzl is None else nzl
        if p["cycle"]:
            self._initial_state_from_files(z0, nzl)
            return
        rR, rB = initial_densities_rk3d(self.isDomain, min(self.nbuf, self.zDomain // 4), p["rho0R"], p["rho0B"])
        self.fluidsRhoR, self.fluidsRhoB = rR[z0:z0 + nzl], rB[z0:z0 + nzl]

    def _initial_state_from_files(self, z0, nzl):
        """[CyclesSetup] IsCycle = 'yes': RKD2Q9.py:491-559 with planes in the place of rows"""
        p = self.par
        dom = self.isDomain[z0:z0 + nzl]

        def find(stem):
            f = find_result_file(self.initial_dir, stem)
            if f is None:
                raise config.ConfigError("IsCycle = 'yes': no %s.h5/.npz in %s" % (stem, self.initial_dir))
            return f

        def need(path, key, tail=()):
            try:
                a = np.array(read_planes(path, key, z0, nzl), dtype=np.float64)
            except KeyError:
                raise config.ConfigError("IsCycle = 'yes': dataset %s missing in %s" % (key, path))
            if a.shape != dom.shape + tuple(tail):
                raise config.ConfigError("IsCycle = 'yes': %s has shape %s, the planes %d..%d of the domain are %s" % (key, a.shape, z0, z0 + nzl - 1, dom.shape + tuple(tail)))
            return a

        zz = np.arange(z0, z0 + nzl)
        if not self._is_image:
            # last record of the previous run; the top 20 planes refilled with blue; f = f_eq(rho, u) (:492-508)
            path, k = find("SimulationResultsRK3D"), p["last_step"]
            self.fluidsRhoR = need(path, "/FluidMacro/FluidDensityRin%d" % k)
            self.fluidsRhoB = need(path, "/FluidMacro/FluidDensityBin%d" % k)
            self.physicalVX, self.physicalVY, self.physicalVZ = (need(path, "/FluidVelocity/FluidVelocity%sAt%d" % (ax, k)) for ax in "XYZ")
            top = zz >= self.zDomain - 20
            self.fluidsRhoR[top] = 0.0
            self.fluidsRhoB[top] = p["rho0B"]
            for a in (self.fluidsRhoR, self.fluidsRhoB, self.physicalVX, self.physicalVY, self.physicalVZ):
                a[dom != 1] = 0.0
        else:
            # cycleInitialRK3D: populations taken over, colours swapped in the top buffer planes (:532-556)
            path = find("cycleInitialRK3D")
            rR, rB = need(path, "/FluidMacro/FluidDensityR"), need(path, "/FluidMacro/FluidDensityB")
            fR, fB = need(path, "/FluidPDF/FluidPDFR", (19,)), need(path, "/FluidPDF/FluidPDFB", (19,))
            top = zz >= self.zDomain - self.nbuf
            self.fluidsRhoR, self.fluidsRhoB = np.where(top[:, None, None], rB, rR), np.where(top[:, None, None], rR, rB)
            self.fluidPDFR, self.fluidPDFB = np.where(top[:, None, None, None], fB, fR), np.where(top[:, None, None, None], fR, fB)
            self.physicalVX, self.physicalVY, self.physicalVZ = (need(path, "/FluidVelocity/FluidVelocity%s" % ax) for ax in "XYZ")

    def _upload_initial_state(self, slab):
        if self.fluidPDFR is not None:
            slab.set_pdf(self.fluidPDFR, self.fluidPDFB, post_collision=True)       # what the reference's arrays hold when recorded
        elif self.physicalVX is not None:
            slab.set_macro(self.fluidsRhoR, self.fluidsRhoB, self.physicalVX, self.physicalVY, self.physicalVZ)
        else:
            slab.set_density(self.fluidsRhoR, self.fluidsRhoB)

    def _distributed(self):
        try:
            import torch.distributed as dist
            return dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1
        except ImportError:
            return False

    # ---- exact checkpoints (lbmpm_rk3d_get_state / set_state)
    def _load_checkpoint(self, slab, z0, nzl):
        path = self.restart_from
        info = [int(v) for v in read_planes(path, "/Checkpoint/Info")]
        S, steps, post, nz, ny, nx, records = info[:7]
        if (nz, ny, nx) != self.isDomain.shape:
            raise config.ConfigError("restart_from: the checkpoint holds a %s lattice, the domain is %s" % ((nz, ny, nx), self.isDomain.shape))
        slab.set_state(read_planes(path, "/Checkpoint/State", z0, nzl), steps, bool(post))
        return steps, records

    def checkpoint(self, path=None):
        """the solver's stored state + step and record counters -> <output_dir>/CheckpointRK3D (collective in a distributed run: rank 0
        writes the stacked state).  restart_from=<that file> continues bit for bit."""
        slab = self._slab
        st, info = slab.get_state()
        st = self._gather(st)
        if st is None:
            return None
        directory, name = (os.path.dirname(path), os.path.splitext(os.path.basename(path))[0]) if path else (self.output_dir, "CheckpointRK3D")
        out = ResultFile(directory, name, (("Checkpoint", "ExactState"),))
        out.write("Checkpoint", "State", st)
        out.write("Checkpoint", "Info", np.array([info["doubles_per_cell"], info["steps"], int(info["post_collision"]), self.zDomain, self.yDomain,
                                                    self.xDomain, self.records], dtype=np.int64))
        return out.path

    def write_cycle_initial(self, directory=None):
        """the current state as <directory>/cycleInitialRK3D (the file the image branch of IsCycle = 'yes' starts from)"""
        slab = self._slab
        self._observe()
        fields = dict(R=slab.get("rhoR"), B=slab.get("rhoB"), X=slab.get("vx"), Y=slab.get("vy"), Z=slab.get("vz"))
        fR, fB = slab.get_pdf()
        got = {k: self._gather(a) for k, a in dict(fields, PR=fR, PB=fB).items()}
        if got["R"] is None:
            return None
        out = ResultFile(directory or self.initial_dir, "cycleInitialRK3D", GROUPS)
        out.write("FluidMacro", "FluidDensityR", got["R"]); out.write("FluidMacro", "FluidDensityB", got["B"])
        out.write("FluidPDF", "FluidPDFR", got["PR"]); out.write("FluidPDF", "FluidPDFB", got["PB"])
        for ax in "XYZ":
            out.write("FluidVelocity", "FluidVelocity" + ax, got[ax])
        return out.path

    def runRKColorGradient3D(self, progress=None):
        p = self.par
        self.initializeDomainBorder()
        par = {k: p[k] for k in PARAM_KEYS}
        name, rank = "SimulationResultsRK3D", 0
        self._gather = lambda a: a
        whole_arrays = False          # distributed 3-D CSF: the slabs carry images of their neighbours' planes and cut them out of the undivided arrays
        if p["tension_type"] == "CSF":
            whole_arrays = self._distributed()
            slab = sim = _CSFSlab(self.isDomain, p, self.device, self.csf_bulk_epsilon, distributed=whole_arrays)
            step, observe = slab.step_single, (lambda: None)
            self.z0, self.nzl = 0, self.zDomain
            if whole_arrays:
                import torch.distributed as dist
                rank = dist.get_rank()
                self.z0, self.nzl = sim.solver.z0, sim.solver.nzl
                if self.gather_records:
                    self._gather = sim.solver.gather
                else:
                    name += "_rank%d" % rank
        elif self._distributed():
            import torch.distributed as dist
            rank = dist.get_rank()
            sim = RK3DDistributed(self.isDomain, par, device=self.device)
            if getattr(self, "calibrate_partition", self.timeSteps >= 1000):
                # long runs: one measured re-cut of the slabs (rk3d.RK3DDistributed.calibrated_plane_cost) from the plain initial
                # state -- the rank that holds the colour interface otherwise runs ~10 % behind the others
                rR, rB = initial_densities_rk3d(self.isDomain, min(self.nbuf, self.zDomain // 4), p["rho0R"], p["rho0B"])
                sim.set_density(rR, rB)
                cost = sim.calibrated_plane_cost(8)
                sim.close()
                sim = RK3DDistributed(self.isDomain, par, device=self.device, plane_cost=cost)
            step, observe, slab = sim.step, sim.observe, sim.slab
            self.z0, self.nzl = sim.z0, sim.nzl
            if self.gather_records:
                self._gather = sim.gather
            else:
                name += "_rank%d" % rank
        else:
            slab = sim = RK3DSlab(self.isDomain, 0, self.zDomain, par, self.device)
            step = slab.step_single
            observe = lambda: slab.phase_field(diagnostics=True)
            self.z0, self.nzl = 0, self.zDomain
        self._slab, self._observe = slab, observe
        done = 0
        z0, nzl = (0, self.zDomain) if whole_arrays else (self.z0, self.nzl)
        if self.restart_from:
            done, self.records = self._load_checkpoint(slab, z0, nzl)
        else:
            self.initializeDomainCondition(z0, nzl)
            self._upload_initial_state(slab)
        writes = rank == 0 or not (self._distributed() and self.gather_records)
        out = ResultFile(self.output_dir, name, GROUPS) if writes else None
        self.result_path = out.path if out else None
        # distributed: every rank checks its own slab, the verdict is collective (all ranks raise together, none is left in an exchange)
        self._guard = RecordGuard("rk3d", slab.num_fluid_nodes, getattr(self, "nan_guard", "raise"), collective=self._distributed(), device=self.device)
        while done < self.timeSteps:
            self._step_now = done
            if done % self.timeInterval == 0:      # (a restarted run records its first step again: the counters of the checkpoint precede it)
                observe()
                self._record(slab, out)
            n = min(self.timeInterval - done % self.timeInterval, self.timeSteps - done)
            if self.checkpoint_every > 0:
                n = min(n, self.checkpoint_every - done % self.checkpoint_every)
            step(n)
            done += n
            if self.checkpoint_every > 0 and done % self.checkpoint_every == 0 and done < self.timeSteps:
                slab.sync()
                self.checkpoint_path = self.checkpoint()
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
        guard = getattr(self, "_guard", None)
        if guard:
            guard(k, getattr(self, "_step_now", 0), dict(rhoR=self.fluidsRhoR, rhoB=self.fluidsRhoB, vx=self.physicalVX, vy=self.physicalVY, vz=self.physicalVZ),
                  dict(massR=float(self.fluidsRhoR.sum()), massB=float(self.fluidsRhoB.sum())))
        items = [("FluidMacro", "FluidDensityRin%g" % k, self.fluidsRhoR), ("FluidMacro", "FluidDensityBin%g" % k, self.fluidsRhoB)]
        items += [("FluidVelocity", "FluidVelocity%sAt%g" % (axis, k), a) for axis, a in (("X", self.physicalVX), ("Y", self.physicalVY), ("Z", self.physicalVZ))]
        if self.record_pdf:
            self.fluidPDFR, self.fluidPDFB = slab.get_pdf()
            items += [("FluidPDF", "FluidPDFBat%g" % k, self.fluidPDFB), ("FluidPDF", "FluidPDFRat%g" % k, self.fluidPDFR)]
        for group, name, a in items:
            a = self._gather(a)         # (collective: every rank passes through, rank 0 -- or everyone, ungathered -- writes)
            if out is not None and a is not None:
                out.write(group, name, a)
        self.records += 1
