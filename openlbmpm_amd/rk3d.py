"""D3Q19 colour-gradient solver (perturbation operator) -- Python face of lbmpm_rk3d_*
(include/lbmpm.h) with z-slab decomposition over torch.distributed (RCCL) or over virtual
ranks inside one process.  All lattice arithmetic happens in liblbmpm_hip.so."""
import ctypes as C

import numpy as np

from . import _lib
from ._lib import F64P, U8P, RK3DConfig, check
from .slab import partition_z, partition_z_balanced, neighbour_exchange, local_exchange, DeviceBuffer

FIELDS = dict(phi=0, rhoR=1, rhoB=2, vx=3, vy=4, vz=5)
BUF = dict(f_send_up=0, f_send_down=1, f_recv_below=2, f_recv_above=3,
           phi_send_up=4, phi_send_down=5, phi_recv_below=6, phi_recv_above=7)

# names follow IniFiles/RKtwophasesetup3D.ini
DEFAULT_PARAMS = dict(AkR=7.0e-3, AkB=7.0e-3, beta=1.0, tauR=1.0, tauB=1.0, SolidRhoR=0.7, SolidRhoB=0.0,
                      velocityZR=0.0, velocityZB=-1.0e-4, densityRL=1.0e-8, densityBL=1.0, relax="SRT",
                      recolor_axis=0.0, recolor_diag=0.0,      # recolor_*: lbmpm_rk3d_config (0 = the model's own weights)
                      inlet="Neumann", densityRH=1.0e-8, densityBH=1.0,      # BoundaryTypeInlet 'Dirichlet': pressure inlet per colour
                      outlet="Dirichlet")                                    # BoundaryTypeOutlet 'Convective': planes 0 .. 2 copy plane 3


class RK3DSlab:
    """One slab: planes [z0, z0+nzl) of a global [nz, ny, nx] lattice."""

    def __init__(self, is_domain_global, z0, nzl, params=None, device=0):
        L = _lib.lib()
        p = dict(DEFAULT_PARAMS); p.update(params or {})
        unknown = set(p) - set(DEFAULT_PARAMS)
        if unknown:
            raise KeyError("unknown RK3D parameters: %s" % sorted(unknown))
        dom = np.ascontiguousarray(is_domain_global, dtype=np.uint8)
        if dom.ndim != 3:
            raise TypeError("is_domain must be a 3-D array [nz, ny, nx]")
        self.nz, self.ny, self.nx = dom.shape
        self.z0, self.nzl, self.device = int(z0), int(nzl), int(device)
        halo = np.zeros((nzl + 2, self.ny, self.nx), dtype=np.uint8)
        lo, hi = max(z0 - 1, 0), min(z0 + nzl + 1, self.nz)
        halo[lo - (z0 - 1):hi - (z0 - 1)] = dom[lo:hi]
        self.is_domain = np.ascontiguousarray(dom[z0:z0 + nzl])
        cfg = RK3DConfig()
        cfg.nx, cfg.ny, cfg.nz_local, cfg.nz_global, cfg.z_offset = self.nx, self.ny, nzl, self.nz, z0
        cfg.ak_r, cfg.ak_b, cfg.beta = p["AkR"], p["AkB"], p["beta"]
        cfg.tau_r, cfg.tau_b = p["tauR"], p["tauB"]
        cfg.solid_phi = (p["SolidRhoR"] - p["SolidRhoB"]) / (p["SolidRhoR"] + p["SolidRhoB"])
        cfg.inlet_vz_r, cfg.inlet_vz_b = p["velocityZR"], p["velocityZB"]
        cfg.outlet_rho_r, cfg.outlet_rho_b = p["densityRL"], p["densityBL"]
        if p["relax"] not in ("SRT", "MRT"):
            raise ValueError("RelaxationType must be 'SRT' or 'MRT'")
        cfg.device, cfg.variant, cfg.relaxation = int(device), 0, int(p["relax"] == "MRT")
        cfg.recolor_axis, cfg.recolor_diag = float(p["recolor_axis"]), float(p["recolor_diag"])
        if p["inlet"] not in ("Neumann", "Dirichlet"):
            raise ValueError("BoundaryTypeInlet must be 'Neumann' or 'Dirichlet'")
        cfg.inlet_type, cfg.inlet_rho_r, cfg.inlet_rho_b = int(p["inlet"] == "Dirichlet"), float(p["densityRH"]), float(p["densityBH"])
        if p["outlet"] not in ("Dirichlet", "Convective"):
            raise ValueError("BoundaryTypeOutlet must be 'Dirichlet' or 'Convective'")
        cfg.outlet_type = int(p["outlet"] == "Convective")
        self._h = C.c_void_p()
        check(L.lbmpm_rk3d_create(C.byref(cfg), halo.ctypes.data_as(U8P), C.byref(self._h)), "lbmpm_rk3d_create")
        self._L = L
        self._tensors = {}

    def close(self):
        if getattr(self, "_h", None) is not None and self._h:
            self._L.lbmpm_rk3d_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def set_density(self, rhoR, rhoB):
        a = np.ascontiguousarray(rhoR, dtype=np.float64); b = np.ascontiguousarray(rhoB, dtype=np.float64)
        if a.shape != (self.nzl, self.ny, self.nx) or b.shape != a.shape:
            raise TypeError("density arrays must have shape %s" % ((self.nzl, self.ny, self.nx),))
        check(self._L.lbmpm_rk3d_set_density(self._h, a.ctypes.data_as(F64P), b.ctypes.data_as(F64P)), "set_density")

    # ---- state in and out (include/lbmpm.h: "State in and out"); arrays cover the slab's own planes
    def _plane_array(self, a, name, last=()):
        if a is None:
            return None, None
        a = np.ascontiguousarray(a, dtype=np.float64)
        if a.shape != (self.nzl, self.ny, self.nx) + tuple(last):
            raise TypeError("%s must have shape %s" % (name, (self.nzl, self.ny, self.nx) + tuple(last)))
        return a, a.ctypes.data_as(F64P)

    def set_macro(self, rhoR, rhoB, vx=None, vy=None, vz=None):
        """f = rho w (1 + 3 e.u + 4.5 (e.u)^2 - 1.5 u^2) per colour (RKD2Q9.py:577-601 in 3-D); the state is the streamed lattice"""
        keep = [self._plane_array(a, n) for a, n in ((rhoR, "rhoR"), (rhoB, "rhoB"), (vx, "vx"), (vy, "vy"), (vz, "vz"))]
        check(self._L.lbmpm_rk3d_set_macro(self._h, *[k[1] for k in keep]), "lbmpm_rk3d_set_macro")

    def set_pdf(self, fR, fB, post_collision=True):
        """populations per colour, [nzl][ny][nx][19] (the 3-D fluidPDFR / fluidPDFB); post_collision: as a time step leaves them"""
        a, pa = self._plane_array(fR, "fR", (19,)); b, pb = self._plane_array(fB, "fB", (19,))
        check(self._L.lbmpm_rk3d_set_pdf(self._h, pa, pb, int(bool(post_collision))), "lbmpm_rk3d_set_pdf")

    def get_pdf(self):
        fR = np.empty((self.nzl, self.ny, self.nx, 19)); fB = np.empty_like(fR)
        check(self._L.lbmpm_rk3d_get_pdf(self._h, fR.ctypes.data_as(F64P), fB.ctypes.data_as(F64P)), "lbmpm_rk3d_get_pdf")
        return fR, fB

    def state_info(self):
        out = (C.c_int64 * 3)()
        check(self._L.lbmpm_rk3d_state_info(self._h, out), "lbmpm_rk3d_state_info")
        return dict(doubles_per_cell=int(out[0]), steps=int(out[1]), post_collision=bool(out[2]))

    def get_state(self):
        """(state [nzl][ny][nx][S], info): what a cell stores, as stored -- the argument of set_state for a bit-exact continuation"""
        info = self.state_info()
        st = np.empty((self.nzl, self.ny, self.nx, info["doubles_per_cell"]))
        check(self._L.lbmpm_rk3d_get_state(self._h, st.ctypes.data_as(F64P)), "lbmpm_rk3d_get_state")
        return st, info

    def set_state(self, state, steps=0, post_collision=True):
        st = np.ascontiguousarray(state, dtype=np.float64)
        if st.ndim != 4 or st.shape[:3] != (self.nzl, self.ny, self.nx):
            raise TypeError("state must have shape %s + (S,)" % ((self.nzl, self.ny, self.nx),))
        check(self._L.lbmpm_rk3d_set_state(self._h, st.ctypes.data_as(F64P), st.shape[3], int(steps), int(bool(post_collision))), "lbmpm_rk3d_set_state")

    @property
    def post_collision(self):
        """the stored populations are those a time step left (the next step streams them); False right after set_density / set_macro"""
        return self.state_info()["post_collision"]

    def use_torch_stream(self, stream):
        """Run this slab's kernels on a torch.cuda.Stream so that they are ordered with torch's
        copies / RCCL operations issued under `with torch.cuda.stream(stream)`.  (Handle 0, the
        legacy default stream, cannot be expressed through the C ABI: pass a real stream.)"""
        if int(stream.cuda_stream) == 0:
            raise ValueError("use a dedicated torch.cuda.Stream, not the default stream")
        check(self._L.lbmpm_rk3d_set_stream(self._h, C.c_void_p(int(stream.cuda_stream))), "set_stream")

    def buffer(self, name):
        """torch view (zero copy) of a halo buffer; None for a buffer this storage does not use (the q23 storage moves the
        halo planes' phase field inside the one face message, its phi buffers have no bytes)"""
        if name not in self._tensors:
            ptr, n = C.c_void_p(), C.c_int64(0)
            check(self._L.lbmpm_rk3d_buffer(self._h, BUF[name], C.byref(ptr), C.byref(n)), "lbmpm_rk3d_buffer")
            self._tensors[name] = DeviceBuffer(ptr.value, n.value).tensor("cuda:%d" % self.device) if n.value > 0 else None
        return self._tensors[name]

    def pack(self):
        check(self._L.lbmpm_rk3d_pack_halo(self._h), "pack_halo")

    def unpack(self, have_below, have_above):
        check(self._L.lbmpm_rk3d_unpack_halo(self._h, int(have_below), int(have_above)), "unpack_halo")

    def phase_field(self, diagnostics=False):
        check(self._L.lbmpm_rk3d_phase_field(self._h, 1 if diagnostics else 0), "phase_field")

    def collide(self):
        check(self._L.lbmpm_rk3d_collide(self._h), "collide")

    def collide_interior(self):
        """start the planes that do not depend on the neighbours on the slab's second stream"""
        check(self._L.lbmpm_rk3d_collide_interior(self._h), "collide_interior")

    def collide_boundary(self):
        check(self._L.lbmpm_rk3d_collide_boundary(self._h), "collide_boundary")

    def step_slab(self, n, has_below, has_above, exchange=None, timed=False):
        """n whole time steps behind one C call (lbmpm_rk3d_step_slab); `exchange(what)` enqueues the transfer of the
        population (0) / phase-field (1) halo buffers on the slab's stream"""
        from ._lib import EXCHANGE_FN
        err = []

        def _cb(_user, what):
            try:
                exchange(int(what))
                return 0
            except BaseException as e:          # an exception must not unwind through the C frames
                err.append(e)
                return 1
        cb = EXCHANGE_FN(_cb) if exchange is not None else None
        rc = self._L.lbmpm_rk3d_step_slab(self._h, int(n), int(bool(has_below)), int(bool(has_above)),
                                          C.cast(cb, C.c_void_p) if cb is not None else None, None, int(bool(timed)))
        if err:
            raise err[0]
        check(rc, "lbmpm_rk3d_step_slab")

    # ---- the exchange's transport inside the library (include/lbmpm.h: LBMPM_TRANSPORT_*)
    def ipc_init(self):
        """allocate this slab's landing area; returns the bytes its neighbours need (lbmpm_rk3d_ipc_init)"""
        blob = C.create_string_buffer(_lib.IPC_BLOB_BYTES)
        check(self._L.lbmpm_rk3d_ipc_init(self._h, blob), "lbmpm_rk3d_ipc_init")
        return blob.raw

    def ipc_connect(self, blob_below, blob_above):
        """map the neighbours' landing areas (their ipc_init bytes; None where the slab has no neighbour)"""
        keep = [C.create_string_buffer(b, _lib.IPC_BLOB_BYTES) if b is not None else None for b in (blob_below, blob_above)]
        check(self._L.lbmpm_rk3d_ipc_connect(self._h, keep[0], keep[1]), "lbmpm_rk3d_ipc_connect")

    @staticmethod
    def rccl_unique_id(librccl_path=None):
        idb = C.create_string_buffer(_lib.RCCL_ID_BYTES)
        check(_lib.lib().lbmpm_rccl_unique_id(idb, librccl_path.encode() if librccl_path else None), "lbmpm_rccl_unique_id")
        return idb.raw

    def rccl_connect(self, unique_id, rank, nranks, librccl_path=None):
        """collective over the ranks of the run: ncclCommInitRank inside the library"""
        idb = C.create_string_buffer(unique_id, _lib.RCCL_ID_BYTES)
        check(self._L.lbmpm_rk3d_rccl_connect(self._h, idb, int(rank), int(nranks), librccl_path.encode() if librccl_path else None),
              "lbmpm_rk3d_rccl_connect")

    def transport_disconnect(self):
        check(self._L.lbmpm_rk3d_transport_disconnect(self._h), "lbmpm_rk3d_transport_disconnect")

    @property
    def transport(self):
        """'callback' (none connected: the caller's exchange function), 'ipc' or 'rccl'; with ipc also how the flags travel"""
        v = C.c_int(0)
        k = self._L.lbmpm_rk3d_transport_kind(self._h, C.byref(v))
        return {0: "callback", 1: "ipc (copy engine + %s)" % ("stream value operations" if v.value else "one-lane flag kernels"), 2: "rccl"}[k]

    def halo_exchange(self):
        check(self._L.lbmpm_rk3d_halo_exchange(self._h), "lbmpm_rk3d_halo_exchange")

    def transport_probe(self, rounds=6):
        check(self._L.lbmpm_rk3d_transport_probe(self._h, int(rounds)), "lbmpm_rk3d_transport_probe")

    def transport_probe_result(self):
        v = C.c_int64(-1)
        check(self._L.lbmpm_rk3d_transport_probe_result(self._h, C.byref(v)), "lbmpm_rk3d_transport_probe_result")
        return int(v.value)

    def ipc_release_waits(self):
        check(self._L.lbmpm_rk3d_ipc_release_waits(self._h), "lbmpm_rk3d_ipc_release_waits")

    def slab_timing(self):
        """dict of average ms over the timed steps of the last step_slab(..., timed=True)"""
        out = (C.c_double * 5)()
        check(self._L.lbmpm_rk3d_slab_timing(self._h, out), "lbmpm_rk3d_slab_timing")
        return dict(step_ms=out[0], interior_ms=out[1], exchange_chain_ms=out[2], boundary_ms=out[3], steps=int(out[4]))

    def step_single(self, n):
        check(self._L.lbmpm_rk3d_step(self._h, int(n)), "lbmpm_rk3d_step")

    def step_timed(self, n):
        a, b = C.c_double(0), C.c_double(0)
        check(self._L.lbmpm_rk3d_step_timed(self._h, int(n), C.byref(a), C.byref(b)), "step_timed")
        return a.value, b.value

    def sync(self, deadline_s=None):
        """wait for the slab's streams; with deadline_s the library's watchdog (lbmpm_rk3d_sync_deadline): LbmpmError with status
        LBMPM_ERR_TIMEOUT (-6) when a neighbour's face message does not arrive in time -- the waits are released, the state is void"""
        if deadline_s is None:
            check(self._L.lbmpm_rk3d_sync(self._h), "sync")
        else:
            check(self._L.lbmpm_rk3d_sync_deadline(self._h, float(deadline_s)), "lbmpm_rk3d_sync_deadline")

    def get(self, name):
        out = np.empty((self.nzl, self.ny, self.nx), dtype=np.float64)
        check(self._L.lbmpm_rk3d_get_field(self._h, FIELDS[name], out.ctypes.data_as(F64P)), "get_field(%s)" % name)
        return out

    @property
    def num_fluid_nodes(self):
        return int(self._L.lbmpm_rk3d_num_fluid_nodes(self._h))

    @property
    def device_bytes(self):
        """device memory held by this context"""
        return int(self._L.lbmpm_rk3d_device_bytes(self._h))

    def debug_plane(self, comp, zl):
        """development aid: stored component `comp` (0..18 populations, 19..22 record, 23 phase-field array, 24 row flags) of the
        local plane zl (0 and nzl + 1: the halo planes) of the current state"""
        n = self.ny * ((self.nx + 63) // 64 if comp == 24 else self.nx)
        out = np.zeros(n, dtype=np.float64)
        check(self._L.lbmpm_rk3d_debug_plane(self._h, int(comp), int(zl), out.ctypes.data_as(F64P)), "lbmpm_rk3d_debug_plane")
        return out.reshape(self.ny, -1)

    def storage_info(self):
        """what the storage keeps and moves, by its own count (lbmpm_rk3d_storage_info)"""
        out = (C.c_int64 * 4)()
        check(self._L.lbmpm_rk3d_storage_info(self._h, out), "lbmpm_rk3d_storage_info")
        return dict(doubles_per_cell=int(out[0]), fluid_cells=int(out[1]), cells_in_flagged_rows=int(out[2]), bytes_per_step=int(out[3]))

    @property
    def steps_done(self):
        return int(self._L.lbmpm_rk3d_steps_done(self._h))

    @property
    def dominant_kernel(self):
        return self._L.lbmpm_rk3d_dominant_kernel(self._h).decode()

    @property
    def one_exchange(self):
        """True for the q23 storage: one face message per step carries populations, records, row flags and the class sums the
        neighbour needs for the phase field of its halo plane (csrc/rk3dq.h); it is needed before the first step too"""
        return self.dominant_kernel == "rk3dq_fused"


def _torch_librccl():
    """the librccl that ships inside the torch wheel (the one torch.distributed's nccl backend uses), or None: the system's"""
    import os
    try:
        import torch
        p = os.path.join(os.path.dirname(torch.__file__), "lib", "librccl.so")
        return p if os.path.exists(p) else None
    except ImportError:
        return None


class RK3DCluster:
    """k slabs ('virtual ranks') in ONE process on one GPU, halos moved by device copies.
    Exists to prove that the slab-decomposed time step equals the single-domain one."""

    def __init__(self, is_domain_global, k, params=None, device=0):
        nz = is_domain_global.shape[0]
        self.parts = partition_z(nz, k)
        import torch
        self.slabs = [RK3DSlab(is_domain_global, z0, n, params, device) for z0, n in self.parts]
        self.stream = torch.cuda.Stream(device)
        self._torch = torch
        for s in self.slabs:
            s.use_torch_stream(self.stream)
        self.k = k

    def set_density(self, rhoR, rhoB):
        for s, (z0, n) in zip(self.slabs, self.parts):
            s.set_density(rhoR[z0:z0 + n], rhoB[z0:z0 + n])

    def set_macro(self, rhoR, rhoB, vx=None, vy=None, vz=None):
        cut = lambda a, z0, n: None if a is None else a[z0:z0 + n]
        for s, (z0, n) in zip(self.slabs, self.parts):
            s.set_macro(*[cut(a, z0, n) for a in (rhoR, rhoB, vx, vy, vz)])

    def set_pdf(self, fR, fB, post_collision=True):
        for s, (z0, n) in zip(self.slabs, self.parts):
            s.set_pdf(fR[z0:z0 + n], fB[z0:z0 + n], post_collision)

    def get_pdf(self):
        got = [s.get_pdf() for s in self.slabs]
        return np.concatenate([g[0] for g in got], axis=0), np.concatenate([g[1] for g in got], axis=0)

    def get_state(self):
        got = [s.get_state() for s in self.slabs]
        return np.concatenate([g[0] for g in got], axis=0), got[0][1]

    def set_state(self, state, steps=0, post_collision=True):
        for s, (z0, n) in zip(self.slabs, self.parts):
            s.set_state(state[z0:z0 + n], steps, post_collision)

    def _exchange(self, kind):
        S = self.slabs
        if S[0].buffer(kind + "_send_up") is None:
            return
        local_exchange([s.buffer(kind + "_send_up") for s in S], [s.buffer(kind + "_send_down") for s in S],
                       [s.buffer(kind + "_recv_below") for s in S], [s.buffer(kind + "_recv_above") for s in S])

    def _halo_f(self):
        for s in self.slabs:
            s.pack()
        self._exchange("f")
        for r, s in enumerate(self.slabs):
            s.unpack(r > 0, r + 1 < self.k)

    def step(self, n):
        with self._torch.cuda.stream(self.stream):
            for _ in range(int(n)):
                for s in self.slabs:
                    s.collide_interior()
                if self.slabs[0].post_collision or self.slabs[0].one_exchange:
                    self._halo_f()
                for s in self.slabs:
                    s.phase_field()
                self._exchange("phi")
                for s in self.slabs:
                    s.collide_boundary()

    def observe(self):
        """rho, u, phi of the streamed + boundary-corrected lattice (what the next step starts from)"""
        with self._torch.cuda.stream(self.stream):
            if self.slabs[0].post_collision or self.slabs[0].one_exchange:
                self._halo_f()
            for s in self.slabs:
                s.phase_field(diagnostics=True)

    def get(self, name):
        return np.concatenate([s.get(name) for s in self.slabs], axis=0)

    def close(self):
        for s in self.slabs:
            s.close()


class RK3DDistributed:
    """One slab per process/GPU.  The face messages travel over a transport INSIDE the library where one connects -- 'ipc' (landing
    areas mapped with hipIpcOpenMemHandle, one copy-engine transfer + a stream value operation per message: no CU, no host work per
    step) or 'rccl' (ncclSend / ncclRecv of a communicator the library makes itself) -- else through the 'callback': torch.distributed
    P2P enqueued from Python once per step.  torch.distributed is used for set-up (handles, unique id, agreement) either way."""

    def __init__(self, is_domain_global, params=None, device=0, group=None, balance=True, plane_cost=None, transport=None):
        """plane_cost: measured cost per lattice plane (`calibrated_plane_cost` of an earlier instance); the cuts then equalise it
        instead of the fluid cells.
        transport: 'auto' (default; environment LBMPM_TRANSPORT overrides) | 'ipc' | 'rccl' | 'callback'.  auto = ipc where every
        rank's set-up self-test passes within its deadline, else the callback; a named transport that fails raises on every rank."""
        import torch.distributed as dist
        self.rank, self.world, self.group = dist.get_rank(group), dist.get_world_size(group), group
        self.parts = self.partition(is_domain_global, self.world, balance, plane_cost)      # every rank's (z0, planes)
        z0, n = self.parts[self.rank]
        self.z0, self.nzl = z0, n
        import os
        import torch
        self._torch = torch
        # every collective of this class and of its callers (NaN verdict, calibration, all_gather_object) runs on the process's
        # CURRENT device under NCCL: make that the slab's GPU, whoever constructed us
        if torch.cuda.is_available():
            torch.cuda.set_device(device)
        self.device = device
        self.slab = RK3DSlab(is_domain_global, z0, n, params, device)
        self.stream = torch.cuda.Stream(device)
        self.slab.use_torch_stream(self.stream)
        want = transport or os.environ.get("LBMPM_TRANSPORT", "auto")
        if want not in ("auto", "ipc", "rccl", "callback"):
            raise ValueError("transport must be 'auto', 'ipc', 'rccl' or 'callback'")
        self.transport_note, self.host_us_per_step = "", 0.0
        # what set-up tried, in order: [{"transport": kind, "ok": bool, "why": text}] -- the diagnosis of a run that ended on another
        # transport than expected (bench.py --gpus N prints it per rank)
        self.transport_log = []
        self.deadline_s = float(os.environ.get("LBMPM_SLAB_DEADLINE_S", "120"))      # steady-state watchdog of sync() / observe()
        if self.world > 1 and want != "callback":
            if not self.slab.one_exchange:
                # (decided by the lattice and the environment, i.e. alike on every rank: no collective needed to agree on it)
                why = "the %s storage has two exchanges per step; the in-library transports carry the one-exchange face message of the 23-value storage" % self.slab.dominant_kernel
                self.transport_log.append(dict(transport=want, ok=False, why=why))
                if want != "auto":
                    raise RuntimeError("transport %r was asked for, but %s" % (want, why))
                self.transport_note = "callback (%s); " % why
            else:
                self._connect(want)

    def _agree(self, ok):
        """True when every rank says ok (MIN over the group; a collective on host objects: works on every backend)"""
        import torch.distributed as dist
        got = [None] * self.world
        dist.all_gather_object(got, bool(ok), group=self.group)
        return all(got)

    def _connect(self, want):
        """Connect the in-library transport; every rank takes the same decision (a transport that works on some ranks only is dropped
        by all).  Each candidate is tried with a self-test under a deadline: patterned face messages each way; a stuck wait (IPC) is
        released by the host, a stuck communicator (RCCL) aborted -- a hang must not stop the job before it has begun.  Every candidate's
        verdict goes to self.transport_log."""
        import torch.distributed as dist
        s, err = self.slab, None
        for kind in (("ipc", "rccl") if want == "auto" and dist.get_backend(self.group) == "nccl" else (("ipc",) if want in ("auto", "ipc") else ("rccl",))):
            # every rank goes through the same collectives in the same order, whatever fails on it: a rank that skipped one would pair
            # its next collective with its neighbours' current one
            ok, err = True, None
            if kind == "ipc":
                try:
                    mine = s.ipc_init()
                except Exception as e:      # noqa: BLE001
                    mine, ok, err = None, False, e
                blobs = [None] * self.world
                dist.all_gather_object(blobs, mine, group=self.group)
                if ok and all(b is not None for b in blobs):
                    try:
                        s.ipc_connect(blobs[self.rank - 1] if self.rank > 0 else None, blobs[self.rank + 1] if self.rank + 1 < self.world else None)
                    except Exception as e:  # noqa: BLE001
                        ok, err = False, e
                elif ok:
                    ok, err = False, "ipc_init failed on rank(s) %s" % [r for r, b in enumerate(blobs) if b is None]
            else:
                box = [None]
                if self.rank == 0:
                    try:
                        box = [s.rccl_unique_id(_torch_librccl())]
                    except Exception as e:  # noqa: BLE001
                        err = e
                dist.broadcast_object_list(box, src=dist.get_global_rank(self.group, 0) if self.group is not None else 0, group=self.group)
                ok = box[0] is not None
                if not ok and err is None:
                    err = "rank 0 could not make a unique id"
                # ipc_open compares the neighbours' message sizes; ncclSend / ncclRecv would silently pair messages of different
                # lengths (different cuts or lattices on two ranks): compare them here, before the blocking collective
                sizes = [None] * self.world
                dist.all_gather_object(sizes, self._face_bytes(), group=self.group)
                for r in range(self.world - 1):
                    if sizes[r]["up"] != sizes[r + 1]["from_below"] or sizes[r + 1]["down"] != sizes[r]["from_above"]:
                        ok, err = False, "ranks %d and %d disagree on the size of the face message across their cut (%s vs %s)" % (r, r + 1, sizes[r], sizes[r + 1])
                if self._agree(ok):         # ncclCommInitRank is a blocking collective: enter it only if every rank will
                    try:
                        s.rccl_connect(box[0], self.rank, self.world, _torch_librccl())
                    except Exception as e:  # noqa: BLE001
                        ok, err = False, e
                else:
                    ok = False
            connected = self._agree(ok)
            tested, why = (self._self_test(kind) if connected else (False, "not tried"))
            if connected and tested:
                self.transport_log.append(dict(transport=kind, ok=True, why="connected; probe of 6 patterned messages each way compared equal on every rank"))
                self.transport_note = "in-library "
                return
            try:
                s.transport_disconnect()
            except Exception:               # noqa: BLE001
                pass
            reason = ("connect: %s" % err) if err else ("another rank could not connect" if not connected else "self-test: %s" % why)
            self.transport_log.append(dict(transport=kind, ok=False, why=str(reason)))
            self.transport_note = "%s did not connect on every rank (%s); " % (kind, reason)
            if want != "auto":
                raise RuntimeError("transport %r could not be connected on every rank: %s" % (kind, reason))

    def _face_bytes(self):
        """sizes of this rank's four face messages (bytes; 0 where it has no neighbour), from its halo buffers"""
        out = {}
        for key, name, there in (("up", "f_send_up", self.rank + 1 < self.world), ("down", "f_send_down", self.rank > 0),
                                 ("from_below", "f_recv_below", self.rank > 0), ("from_above", "f_recv_above", self.rank + 1 < self.world)):
            b = self.slab.buffer(name) if there else None
            out[key] = int(b.numel() * b.element_size()) if b is not None else 0
        return out

    def _self_test(self, kind, deadline_s=20.0):
        """six patterned messages each way between the real neighbours through the new transport (lbmpm_rk3d_transport_probe: every
        landing slot three times, compared on the receiving GPU), under a deadline for EITHER transport: the library's watchdog
        releases a stuck IPC wait / aborts a stuck communicator.  -> (every rank passed, this rank's verdict as text)"""
        s = self.slab
        try:
            with self._torch.cuda.stream(self.stream):
                s.transport_probe(6)
        except Exception as e:              # noqa: BLE001 -- this rank could not even enqueue: tell the others (same collective)
            self._agree(False)
            return False, "could not enqueue the probe: %s" % e
        why = "ok"
        try:
            s.sync(deadline_s=deadline_s)
            bad = s.transport_probe_result()
            mine = bad == 0
            if not mine:
                why = "%d doubles arrived wrong" % bad
        except Exception as e:              # noqa: BLE001 -- the watchdog fired (or the stream failed)
            mine, why = False, str(e)
        every = self._agree(mine)
        return every, (why if not mine else ("ok here, failed on another rank" if not every else "ok"))

    @property
    def transport(self):
        return self.slab.transport

    @staticmethod
    def partition(is_domain_global, world, balance=True, plane_cost=None):
        """z-ranges of the ranks: equal fluid cells per rank (default), equal measured cost (plane_cost, one value per plane), or
        equal planes (balance=False)"""
        if not balance or world == 1:
            return partition_z(is_domain_global.shape[0], world)
        if plane_cost is not None:
            c = np.asarray(plane_cost, dtype=np.float64)
            if c.shape != (is_domain_global.shape[0],) or not np.all(np.isfinite(c)) or c.min() <= 0:
                raise ValueError("plane_cost needs one positive value per plane")
            return partition_z_balanced(np.maximum(1, np.round(c / c.max() * 1e6)).astype(np.int64), world)
        counts = (np.asarray(is_domain_global) == 1).reshape(is_domain_global.shape[0], -1).sum(axis=1)
        return partition_z_balanced(counts, world)

    def calibrated_plane_cost(self, steps=8):
        """Cost per lattice plane as THIS partition runs it: `steps` timed steps, every rank's kernel time per owned plane, gathered --
        the argument `plane_cost` of the next instance.  The march step of rk3dq_fused costs the same at every fluid fraction but more
        where both colours meet, so equal fluid cells leave the rank that holds the interface ~10 % behind the others (DESIGN.md
        section 5); one measured re-cut evens that out for as long as the interface stays within its rank.  Collective; the state is
        advanced by `steps` steps (call set_density again)."""
        import torch.distributed as dist
        self.step(int(steps), timed=True)
        t = self.timing()
        rate = t["step_ms"] / max(self.nzl, 1)          # the whole step (fixed parts included: repeated re-cuts converge on equal step times)
        got = [None] * self.world
        dist.all_gather_object(got, (int(self.z0), int(self.nzl), float(rate)), group=self.group)
        nz = max(z0 + n for z0, n, _ in got)
        cost = np.zeros(nz)
        for z0, n, r in got:
            cost[z0:z0 + n] = r
        if not np.all(cost > 0):
            raise RuntimeError("calibration gave a non-positive plane cost")
        return cost

    def set_density(self, rhoR_global, rhoB_global):
        self.slab.set_density(rhoR_global[self.z0:self.z0 + self.nzl], rhoB_global[self.z0:self.z0 + self.nzl])

    def _mine(self, a):
        return None if a is None else a[self.z0:self.z0 + self.nzl]

    def set_macro(self, rhoR, rhoB, vx=None, vy=None, vz=None):
        """global arrays (every rank takes its planes); see RK3DSlab.set_macro"""
        self.slab.set_macro(*[self._mine(a) for a in (rhoR, rhoB, vx, vy, vz)])

    def set_pdf(self, fR, fB, post_collision=True):
        self.slab.set_pdf(self._mine(fR), self._mine(fB), post_collision)

    def set_state(self, state, steps=0, post_collision=True):
        self.slab.set_state(self._mine(state), steps, post_collision)

    def gather(self, local):
        """this rank's planes of a field / state -> the whole lattice's array on rank 0 (None elsewhere); collective"""
        from .slab import gather_planes
        return gather_planes(local, self.parts, self.rank, self.world, self.group, self.device)

    def _exchange(self, kind):
        s = self.slab
        if s.buffer(kind + "_send_up") is None:
            return
        neighbour_exchange(s.buffer(kind + "_send_up"), s.buffer(kind + "_send_down"),
                           s.buffer(kind + "_recv_below"), s.buffer(kind + "_recv_above"),
                           self.rank, self.world, self.group)

    def _halo_f(self):
        s = self.slab
        if (s.post_collision or s.one_exchange) and self.world > 1:
            if s.transport != "callback":
                s.halo_exchange()
                return
            s.pack()
            self._exchange("f")
            s.unpack(self.rank > 0, self.rank + 1 < self.world)

    def step(self, n, timed=False):
        """n time steps: ONE call into the library (lbmpm_rk3d_step_slab), which runs the interior planes on the slab's
        second stream and calls back twice per step for the two neighbour exchanges (torch.distributed P2P = RCCL
        over xGMI, enqueued on the slab's stream).  timed: per-phase HIP events, read with timing()."""
        import time
        s = self.slab
        own = s.transport != "callback"
        t0 = time.perf_counter()
        with self._torch.cuda.stream(self.stream):
            s.step_slab(n, self.rank > 0, self.rank + 1 < self.world,
                        (lambda what: self._exchange("phi" if what else "f")) if self.world > 1 and not own else None, timed)
        # what the HOST spent enqueueing (the call returns before the GPU has done the work): per step, launches + exchange enqueues
        self.host_us_per_step = (time.perf_counter() - t0) * 1e6 / max(int(n), 1)

    def timing(self):
        t = self.slab.slab_timing()
        if self.slab.one_exchange:      # boundary planes, then the exchange chain beside the interior planes
            t["exchange_exposed_ms"] = max(0.0, t["step_ms"] - t["boundary_ms"] - t["interior_ms"])
            t["schedule"] = "boundary planes -> one face exchange || interior planes"
        else:
            t["exchange_exposed_ms"] = t["step_ms"] - max(t["interior_ms"], t["boundary_ms"])
            t["schedule"] = "interior planes || two exchanges -> boundary planes"
        faces = {}
        for side, there in (("down", self.rank > 0), ("up", self.rank + 1 < self.world)):       # faces that have a neighbour
            if there:
                f = self.slab.buffer("f_send_" + side); ph = self.slab.buffer("phi_send_" + side)
                faces[side] = int(f.numel() * f.element_size() + (ph.numel() * ph.element_size() if ph is not None else 0))
        t["bytes_sent_per_step"] = faces                    # populations (5 x 2 colours, fluid cells of the face plane) + phi plane
        t["bytes_per_face"] = max(faces.values()) if faces else 0
        t["transport"] = (self.transport_note + self.slab.transport) if self.world > 1 else "none (one slab)"
        t["transport_log"] = list(self.transport_log)
        t["host_enqueue_us_per_step"] = round(getattr(self, "host_us_per_step", 0.0), 1)
        return t

    def observe(self):
        with self._torch.cuda.stream(self.stream):
            self._halo_f()
            self.slab.phase_field(diagnostics=True)
        self.sync()

    def sync(self, deadline_s=None):
        """Wait for this rank's work.  With an in-library transport the wait is the library's watchdog (lbmpm_rk3d_sync_deadline,
        `deadline_s` or self.deadline_s = LBMPM_SLAB_DEADLINE_S, default 120 s WITHOUT a completed face exchange -- queued steps that
        drain, slowly or not, keep resetting it): a rank whose neighbour died mid-run raises
        LbmpmError (status -6) after the deadline instead of hanging in hipStreamWaitValue64 for good -- and so does every other
        rank of the broken chain, each at its own deadline."""
        if self.world > 1 and self.slab.transport != "callback":
            self.slab.sync(deadline_s=self.deadline_s if deadline_s is None else deadline_s)
        else:
            self.stream.synchronize()

    def close(self):
        self.slab.close()
