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
    def __init__(self, is_domain, params=None, device=0, diagnostics=False, slab=None):
        """slab = (z0, global_nz): this lattice is a slab of an undivided lattice of global_nz planes -- its planes 2 .. nz-3 are the planes
        z0 .. of that lattice, its two planes at either end images of the neighbouring slabs' edge planes (cut from the undivided lattice,
        wrapping around its ends, like the rest)"""
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
        self.ghost = (GHOST, GHOST) if slab else (0, 0)
        cfg.ghost_lo, cfg.ghost_hi = self.ghost
        if slab:
            cfg.slab_z0, cfg.global_nz = int(slab[0]), int(slab[1])
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

    def stage(self, k):
        """a third of a step (0 phase field, 1 gradient, 2 collision); the face messages go in between (include/lbmpm.h)"""
        check(self._L.lbmpm_rk3dcsf_stage(self._h, int(k)), "lbmpm_rk3dcsf_stage")

    def face_doubles(self, msg, face):
        return int(self._L.lbmpm_rk3dcsf_face_doubles(self._h, int(msg), int(face)))

    def face_doubles_in(self, msg, face):
        return int(self._L.lbmpm_rk3dcsf_face_doubles_in(self._h, int(msg), int(face)))

    def face_pack(self, msg, face, device_ptr):
        check(self._L.lbmpm_rk3dcsf_face_pack(self._h, int(msg), int(face), C.c_void_p(int(device_ptr))), "lbmpm_rk3dcsf_face_pack")

    def face_unpack(self, msg, face, device_ptr):
        check(self._L.lbmpm_rk3dcsf_face_unpack(self._h, int(msg), int(face), C.c_void_p(int(device_ptr))), "lbmpm_rk3dcsf_face_unpack")

    def send_to(self, face, other, msg):
        """same process: this slab's message through `face` into the ghost planes of the slab on the other side"""
        check(self._L.lbmpm_rk3dcsf_face_copy(self._h, int(face), other._h, int(msg)), "lbmpm_rk3dcsf_face_copy")

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


MSG_PDF, MSG_PHI, MSG_NORMAL = 0, 1, 2        # LBMPM_CSF_MSG_*
_AFTER_STAGE = (MSG_PHI, MSG_NORMAL, MSG_PDF)  # the message that follows stage 0, 1, 2
GHOST = 2


def slab_cuts(nz, nslabs, weights=None):
    """z0 of every slab + nz: equal shares of the planes (or of `weights`, one number per plane), at least 4 planes each"""
    nslabs = int(nslabs)
    if nslabs < 1 or nz < 4 * nslabs:
        raise ValueError("%d planes do not make %d slabs of at least 4" % (nz, nslabs))
    w = np.ones(nz) if weights is None else np.asarray(weights, dtype=np.float64)
    acc = np.concatenate([[0.0], np.cumsum(w)])
    cuts = [0]
    for k in range(1, nslabs):
        z = int(np.searchsorted(acc, acc[-1] * k / nslabs))
        cuts.append(min(max(z, cuts[-1] + 4), nz - 4 * (nslabs - k)))
    return cuts + [nz]


class _SlabGeometry:
    """the planes [z0, z1) of a lattice [nz][ny][nx] with the two ghost planes a slab of the CSF model carries at either end (the slabs
    form a ring: the loop wraps z, so the first slab's low neighbour is the last slab)"""

    def __init__(self, nz, z0, z1):
        self.nz, self.z0, self.z1 = int(nz), int(z0), int(z1)
        self.whole = self.z0 == 0 and self.z1 == self.nz
        self.ghost = (0, 0) if self.whole else (GHOST, GHOST)
        self.planes = np.arange(self.z0 - self.ghost[0], self.z1 + self.ghost[1]) % self.nz
        self.slab = None if self.whole else (self.z0, self.nz)

    def cut(self, a):
        return None if a is None else np.ascontiguousarray(np.take(a, self.planes, axis=0))

    def own(self, a):
        return a[self.ghost[0]:a.shape[0] - self.ghost[1]]


class RK3DCSFCluster:
    """The 3-D CSF model cut into slabs along z, every slab a context of its own (devices[k]; all on one GPU: a rehearsal of the
    decomposition, bit-equal to the undivided lattice).  One time step = three stages with a face message after each: phi (two planes),
    n (one plane), the populations crossing the face (lbmpm_rk3dcsf_stage / _face_copy)."""

    def __init__(self, is_domain, params=None, nslabs=2, devices=None, cuts=None, diagnostics=False):
        dom = np.ascontiguousarray(is_domain, dtype=np.uint8)
        self.shape = dom.shape
        self.nz, self.ny, self.nx = dom.shape
        self.cuts = list(cuts) if cuts is not None else slab_cuts(self.nz, nslabs)
        n = len(self.cuts) - 1
        devices = list(devices) if devices is not None else [0] * n
        self.geo = [_SlabGeometry(self.nz, self.cuts[k], self.cuts[k + 1]) for k in range(n)]
        self.slabs = [RK3DCSFSolver(g.cut(dom), params, device=devices[k], diagnostics=diagnostics, slab=g.slab) for k, g in enumerate(self.geo)]
        self.params = self.slabs[0].params

    def close(self):
        for s in self.slabs:
            s.close()

    def set_macro(self, rhoR, rhoB, vx=None, vy=None, vz=None):
        for s, g in zip(self.slabs, self.geo):
            s.set_macro(*[g.cut(None if a is None else np.asarray(a, dtype=np.float64)) for a in (rhoR, rhoB, vx, vy, vz)])

    def set_pdf(self, fR, fB, force=None):
        for s, g in zip(self.slabs, self.geo):
            s.set_pdf(g.cut(np.asarray(fR)), g.cut(np.asarray(fB)), None if force is None else tuple(g.cut(np.asarray(c)) for c in force))

    def _exchange(self, msg):
        n = len(self.slabs)
        for k in range(n):                         # a ring: the last slab's high face is the first slab's low face
            up = self.slabs[(k + 1) % n]
            self.slabs[k].send_to(1, up, msg)
            up.send_to(0, self.slabs[k], msg)

    def step(self, nsteps=1):
        if len(self.slabs) == 1:
            return self.slabs[0].step(nsteps)
        for _ in range(int(nsteps)):
            for stage in range(3):
                for s in self.slabs:
                    s.stage(stage)
                self._exchange(_AFTER_STAGE[stage])

    def sync(self):
        for s in self.slabs:
            s.sync()

    def get(self, name):
        return np.concatenate([g.own(s.get(name)) for s, g in zip(self.slabs, self.geo)], axis=0)

    num_fluid_nodes = property(lambda self: int(self.is_fluid_total))
    steps_done = property(lambda self: self.slabs[0].steps_done)
    dominant_kernel = property(lambda self: self.slabs[0].dominant_kernel)

    @property
    def is_fluid_total(self):
        return sum(int((g.own(s.is_domain) == 1).sum()) for s, g in zip(self.slabs, self.geo))

    @property
    def bulk_cells(self):
        return sum(s.bulk_cells for s in self.slabs)


class RK3DCSFDistributed:
    """One slab of the 3-D CSF model per rank of torch.distributed (launch: one process per GPU).  The face messages travel as device
    tensors under the nccl (= RCCL) backend and through host memory under gloo; three per step and face (phi, n, populations: 2 + 3 +
    10 doubles per cell of a plane), batched per stage."""

    def __init__(self, is_domain, params=None, device=0, cuts=None, diagnostics=False, slab_factory=None):
        """slab_factory: tests/test_slab_cpu.py puts a host stand-in with the library's stage / face calls in the solver's place (attribute
        on_host: its buffers are host tensors) to run this class's orchestration under gloo without a GPU"""
        import torch
        import torch.distributed as dist
        self._torch, self._dist = torch, dist
        self.rank, self.world = dist.get_rank(), dist.get_world_size()
        dom = np.ascontiguousarray(is_domain, dtype=np.uint8)
        self.shape = dom.shape
        self.nz = dom.shape[0]
        # (default: equal shares of the fluid cells -- the same cuts on every rank, they hold the same mask)
        self.cuts = list(cuts) if cuts is not None else slab_cuts(self.nz, self.world, weights=(dom.reshape(self.nz, -1) == 1).sum(axis=1) + 1.0e-3)
        if len(self.cuts) != self.world + 1:
            raise ValueError("one slab per rank: %d cuts for %d ranks" % (len(self.cuts) - 1, self.world))
        self.geo = _SlabGeometry(self.nz, self.cuts[self.rank], self.cuts[self.rank + 1])
        self.z0, self.nzl = self.geo.z0, self.geo.z1 - self.geo.z0
        self.slab = (slab_factory or RK3DCSFSolver)(self.geo.cut(dom), params, device=device, diagnostics=diagnostics, slab=self.geo.slab)
        self.params = self.slab.params
        self._on_device = dist.get_backend() == "nccl"
        dev = self._dev = torch.device("cpu") if getattr(slab_factory, "on_host", False) else torch.device("cuda", int(device))
        self._t_stage, self._t_msg, self._t_count = [0.0] * 3, [0.0] * 3, [0] * 3
        self._buf = {}
        for face in (0, 1):
            if self.geo.ghost[face]:
                for msg in (MSG_PDF, MSG_PHI, MSG_NORMAL):
                    n, m = self.slab.face_doubles(msg, face), self.slab.face_doubles_in(msg, face)
                    self._buf[(msg, face)] = (torch.empty(n, dtype=torch.float64, device=dev), torch.empty(m, dtype=torch.float64, device=dev))
        # every rank cuts its slab out of the same undivided lattice
        import zlib
        mine = torch.tensor([zlib.crc32(dom.tobytes())] + list(dom.shape) + self.cuts, dtype=torch.int64)
        every = [torch.zeros_like(mine) for _ in range(self.world)]
        if self._on_device:
            every = [t.to(dev) for t in every]
            mine = mine.to(dev)
        dist.all_gather(every, mine)
        for r in range(self.world):
            if not bool((every[r] == mine).all()):
                raise ValueError("rank %d holds another lattice or other cuts than rank %d" % (r, self.rank))

    def close(self):
        self.slab.close()

    def _exchange(self, msg):
        torch, dist = self._torch, self._dist
        if self.world == 1:
            return
        lo_peer, hi_peer = (self.rank - 1) % self.world, (self.rank + 1) % self.world
        import time
        t0 = time.perf_counter()
        for face in (0, 1):
            self.slab.face_pack(msg, face, self._buf[(msg, face)][0].data_ptr())
        self.slab.sync()
        t1 = time.perf_counter()
        staged = []
        if self._on_device:
            # (no tags under NCCL: between two ranks the messages pair up in the order posted -- with two ranks both faces join the same
            # pair, so every rank sends low, high and receives high, low: the peer's low-face message is what arrives through my high face)
            ops = [dist.P2POp(dist.isend, self._buf[(msg, 0)][0], lo_peer), dist.P2POp(dist.isend, self._buf[(msg, 1)][0], hi_peer),
                   dist.P2POp(dist.irecv, self._buf[(msg, 1)][1], hi_peer), dist.P2POp(dist.irecv, self._buf[(msg, 0)][1], lo_peer)]
            reqs = dist.batch_isend_irecv(ops)
        else:
            reqs = []
            for face, peer in ((0, lo_peer), (1, hi_peer)):
                out, inn = self._buf[(msg, face)]
                o = out.cpu()
                i = torch.empty(inn.shape, dtype=inn.dtype)
                staged.append((inn, i))
                # tag = the face of the RECEIVER the message enters through
                reqs += [dist.isend(o, peer, tag=1 - face), dist.irecv(i, peer, tag=face)]
        for w in reqs:
            w.wait()
        for inn, i in staged:
            inn.copy_(i)
        if self._dev.type == "cuda":
            # the messages are in the buffers before the library's stream takes them; torch's stream only -- the bulk's collision keeps
            # running on the library's second stream while phi and n travel
            torch.cuda.current_stream(self._dev).synchronize()
        for face in (0, 1):
            self.slab.face_unpack(msg, face, self._buf[(msg, face)][1].data_ptr())
        t2 = time.perf_counter()
        # host clock: [stage's launches on the first stream + pack] until they have run | the message (both faces) until it is unpacked
        k = _AFTER_STAGE.index(msg)
        self._t_stage[k] += t1 - t0
        self._t_msg[k] += t2 - t1
        self._t_count[k] += 1

    def timing(self, reset=True):
        """per step and stage, on this rank's host clock (ms): 'wait_stage' = from the stage's call to the moment its launches on the first
        stream and the pack have run (the bulk's collision on the second stream is not waited for before stage 2), 'message' = from there
        until both faces' messages are unpacked; names of the stages' messages: phi, normal, populations"""
        n = [max(1, c) for c in self._t_count]
        out = dict(steps=int(self._t_count[2]), wait_stage_ms=[round(1e3 * t / c, 4) for t, c in zip(self._t_stage, n)],
                   message_ms=[round(1e3 * t / c, 4) for t, c in zip(self._t_msg, n)], messages=["phi", "normal", "populations"],
                   bytes_per_face=[8 * self.slab.face_doubles(m, 1) for m in _AFTER_STAGE], planes=[self.z0, self.z0 + self.nzl])
        if reset:
            self._t_stage, self._t_msg, self._t_count = [0.0] * 3, [0.0] * 3, [0] * 3
        return out

    def set_macro(self, rhoR, rhoB, vx=None, vy=None, vz=None):
        """the undivided arrays [nz][ny][nx]; every rank takes its planes"""
        self.slab.set_macro(*[self.geo.cut(None if a is None else np.asarray(a, dtype=np.float64)) for a in (rhoR, rhoB, vx, vy, vz)])

    def set_pdf(self, fR, fB, force=None):
        self.slab.set_pdf(self.geo.cut(np.asarray(fR)), self.geo.cut(np.asarray(fB)),
                          None if force is None else tuple(self.geo.cut(np.asarray(c)) for c in force))

    def step(self, nsteps=1):
        if self.world == 1:
            return self.slab.step(nsteps)
        for _ in range(int(nsteps)):
            for stage in range(3):
                self.slab.stage(stage)
                self._exchange(_AFTER_STAGE[stage])       # (its clock starts right after the stage's launches are queued)

    def sync(self):
        self.slab.sync()

    def get(self, name):
        """this rank's own planes"""
        return np.ascontiguousarray(self.geo.own(self.slab.get(name)))

    def gather(self, a):
        """rank 0: the ranks' planes stacked along z (None elsewhere); slab.gather_planes"""
        from .slab import gather_planes
        parts = [(self.cuts[r], self.cuts[r + 1] - self.cuts[r]) for r in range(self.world)]
        return gather_planes(a, parts, self.rank, self.world, device=self._dev.index)

    num_fluid_nodes = property(lambda self: int((self.geo.own(self.slab.is_domain) == 1).sum()))
    steps_done = property(lambda self: self.slab.steps_done)
    dominant_kernel = property(lambda self: self.slab.dominant_kernel)
