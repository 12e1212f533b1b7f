"""CPU tests of the z-slab plumbing (openlbmpm_amd/slab.py): partitioning and the neighbour
exchange over torch.distributed with the gloo backend, world_size 2 and 3 (the N>1 path of the
3-D solver uses exactly this code with backend nccl = RCCL)."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from openlbmpm_amd.slab import partition_z, partition_z_balanced, neighbour_exchange, local_exchange


def test_partition_covers_and_orders():
    for nz, world in ((96, 1), (96, 8), (97, 8), (512, 8), (17, 4)):
        parts = partition_z(nz, world)
        assert parts[0][0] == 0 and sum(n for _, n in parts) == nz
        for (z0, n), (z1, _) in zip(parts, parts[1:]):
            assert z0 + n == z1
        assert min(n for _, n in parts) >= 2 and max(n for _, n in parts) - min(n for _, n in parts) <= 1
    with pytest.raises(ValueError):
        partition_z(7, 4)


def test_local_exchange_matches_definition():
    k, m = 4, 10
    up = [torch.full((m,), 10.0 + r) for r in range(k)]
    dn = [torch.full((m,), 20.0 + r) for r in range(k)]
    below = [torch.zeros(m) for _ in range(k)]
    above = [torch.zeros(m) for _ in range(k)]
    local_exchange(up, dn, below, above)
    for r in range(k):
        assert float(below[r][0]) == (10.0 + r - 1 if r > 0 else 0.0)
        assert float(above[r][0]) == (20.0 + r + 1 if r + 1 < k else 0.0)


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close()
    return p


def _worker(rank, world, port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        m = 64
        # every rank owns a 1-D "lattice" of m*world cells; a step = each cell takes the mean of its
        # two z-neighbours; the slab version needs exactly one halo cell from each side.
        rng = np.random.default_rng(1234)
        full = rng.standard_normal(m * world)
        (z0, n) = partition_z(m * world, world)[rank]
        mine = torch.tensor(full[z0:z0 + n])
        below, above = torch.zeros(1, dtype=torch.float64), torch.zeros(1, dtype=torch.float64)
        for _ in range(3):
            neighbour_exchange(mine[-1:].clone(), mine[:1].clone(), below, above, rank, world)
            ext = torch.cat([below if rank > 0 else mine[:1], mine, above if rank + 1 < world else mine[-1:]])
            mine = 0.5 * (ext[:-2] + ext[2:])
        ref = full.copy()
        for _ in range(3):
            ext = np.concatenate([ref[:1], ref, ref[-1:]])
            ref = 0.5 * (ext[:-2] + ext[2:])
        ok = np.array_equal(mine.numpy(), ref[z0:z0 + n])
        q.put((rank, bool(ok)))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 3])
def test_neighbour_exchange_gloo(world):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    results = [q.get(timeout=120) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert sorted(results) == [(r, True) for r in range(world)]


def _gather_worker(rank, world, port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from openlbmpm_amd.slab import gather_planes, partition_z_balanced
        rng = np.random.default_rng(7)
        full = rng.standard_normal((23, 5, 4, 3))
        parts = partition_z_balanced(rng.integers(1, 9, 23), world)        # uneven cuts
        z0, n = parts[rank]
        ok = True
        for a in (full, full[..., 0], (full[..., 0, 0] > 0).astype(np.uint8)):   # [nz][ny][nx][S], [nz][ny][nx], another dtype
            got = gather_planes(a[z0:z0 + n], parts, rank, world)
            ok = ok and ((got is None) if rank else (got.dtype == a.dtype and np.array_equal(got, a)))
        q.put((rank, bool(ok)))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 8])
def test_gather_planes_stacks_the_ranks_records_on_rank_0(world):
    """the one result file of a distributed run (RKColorGradientD3Q19._record): every rank's planes, point to point to rank 0"""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_gather_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    results = [q.get(timeout=180) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert sorted(results) == [(r, True) for r in range(world)]


def test_balanced_partition_equalises_fluid_cells():
    rng = np.random.default_rng(3)
    w = np.full(512, 170000); w[:10] = 262144; w[-10:] = 262144
    w = w + rng.integers(-3000, 3000, size=512)
    for world in (1, 2, 3, 4, 8):
        parts = partition_z_balanced(w, world)
        assert parts[0][0] == 0 and sum(n for _, n in parts) == 512
        assert all(parts[r][0] + parts[r][1] == parts[r + 1][0] for r in range(world - 1))
        assert all(n >= 2 for _, n in parts)
        loads = [int(w[z0:z0 + n].sum()) for z0, n in parts]
        assert max(loads) <= 1.01 * (w.sum() / world) + w.max()
    # a slab without a fluid cell cannot be created, and one rank failing alone would hang the others in the first
    # exchange: the cut itself refuses, identically on every rank; where an even cut avoids the empty slab it is used
    w = np.zeros(16, dtype=int); w[5] = 100
    with pytest.raises(ValueError):
        partition_z_balanced(w, 8)
    w = np.array([0] * 20 + [5] * 4)
    with pytest.raises(ValueError):
        partition_z_balanced(w, 4)
    w = np.array([1] * 4 + [50] * 4 + [1] * 4)
    parts = partition_z_balanced(w, 3)
    assert all(w[z0:z0 + n].sum() > 0 for z0, n in parts)
    with pytest.raises(ValueError):
        partition_z_balanced(np.ones(7), 4)


def _guard_worker(rank, world, port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from openlbmpm_amd.results import RecordGuard, SimulationDiverged
        g = RecordGuard("rk3d", 100, "raise", collective=True)
        a = np.ones(8)
        g(0, 0, dict(rhoR=a))                     # every slab finite: nobody raises
        bad = a.copy()
        if rank == 1:
            bad[3] = np.nan                       # ONE rank's slab diverges
        try:
            g(1, 8, dict(rhoR=bad))
            q.put((rank, "no exception"))
        except SimulationDiverged as e:
            q.put((rank, "own" if "rhoR holds 1" in str(e) else ("other" if "another rank" in str(e) else str(e))))
        dist.barrier()                            # all ranks are still in step: the next collective completes
    finally:
        dist.destroy_process_group()


def test_record_guard_verdict_is_collective():
    """a NaN in one rank's slab makes EVERY rank raise at that record (one MAX all-reduce of a flag): a rank that raised alone would
    leave its neighbours hanging in the next halo exchange"""
    world = 3
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_guard_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    results = dict(q.get(timeout=120) for _ in range(world))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert results == {0: "other", 1: "own", 2: "other"}


def test_partition_by_measured_plane_cost():
    """RK3DDistributed.partition(plane_cost=...): the cuts equalise the measured cost, not the fluid cells -- a rank whose planes cost
    more gets fewer of them; bad cost arrays are refused"""
    from openlbmpm_amd.rk3d import RK3DDistributed
    dom = np.ones((64, 4, 4), dtype=np.uint8)
    even = RK3DDistributed.partition(dom, 4)
    assert [n for _, n in even] == [16, 16, 16, 16]
    cost = np.ones(64); cost[48:] = 1.5                     # the last quarter is 50 % dearer per plane
    parts = RK3DDistributed.partition(dom, 4, plane_cost=cost)
    assert parts[0][0] == 0 and sum(n for _, n in parts) == 64 and all(parts[i][0] + parts[i][1] == parts[i + 1][0] for i in range(3))
    per_rank = [cost[z0:z0 + n].sum() for z0, n in parts]
    assert parts[-1][1] < 16 and max(per_rank) / min(per_rank) < 1.12
    for bad in (np.ones(63), np.zeros(64), np.full(64, np.nan)):
        with pytest.raises(ValueError):
            RK3DDistributed.partition(dom, 4, plane_cost=bad)


# ---- host logic of the transport selection (openlbmpm_amd/rk3d.py::RK3DDistributed._connect), world size 2 over gloo, no GPU: the
# slab is a stand-in that records the calls and fails where the scenario says so
class _FakeStream:
    def query(self):
        return True

    def synchronize(self):
        pass


class _FakeTorch:
    class cuda:
        @staticmethod
        def stream(_s):
            import contextlib
            return contextlib.nullcontext()


class _FakeSlab:
    one_exchange = True

    def __init__(self, rank, fail):
        self.rank, self.fail, self.kind, self.calls = rank, fail, "callback", []

    def _maybe(self, what):
        self.calls.append(what)
        if self.fail.get(what) == self.rank or self.fail.get(what) == "all":
            raise RuntimeError("%s fails on rank %d" % (what, self.rank))

    def ipc_init(self):
        self._maybe("ipc_init")
        return b"blob-of-rank-%d" % self.rank

    def ipc_connect(self, below, above):
        self._maybe("ipc_connect")
        assert (below, above) == ((None, b"blob-of-rank-1") if self.rank == 0 else (b"blob-of-rank-0", None))
        self.kind = "ipc (copy engine + stream value operations)"

    @staticmethod
    def rccl_unique_id(_path=None):
        return b"U" * 128

    def rccl_connect(self, uid, rank, nranks, _path=None):
        self._maybe("rccl_connect")
        assert uid == b"U" * 128 and rank == self.rank and nranks == 2
        self.kind = "rccl"

    def transport_probe(self, rounds):
        self._maybe("probe")

    def transport_probe_result(self):
        return 3 if self.fail.get("mismatch") in (self.rank, "all") and self.kind.startswith("ipc") else 0

    def transport_disconnect(self):
        self.calls.append("disconnect")
        self.kind = "callback"

    def sync(self, deadline_s=None):
        self.calls.append("sync(%s)" % ("deadline" if deadline_s else "no deadline"))
        if self.fail.get("hang") in (self.rank, "all") and self.kind.startswith(tuple(self.fail.get("hang_kinds", ("ipc", "rccl")))):
            raise RuntimeError("lbmpm_rk3d_sync_deadline: the slab's streams were still busy after %.1f s" % deadline_s)

    class _Buf:
        def __init__(self, n):
            self.n = n

        def numel(self):
            return self.n

        def element_size(self):
            return 8

    def buffer(self, name):
        # the two sides of the cut hold the same message sizes unless the scenario says the ranks disagree
        return self._Buf(1000 + (7 if self.fail.get("sizes") == self.rank and name == "f_send_up" else 0))

    @property
    def transport(self):
        return self.kind


def _connect_worker(rank, world, port, q, want, fail, backend_name):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from openlbmpm_amd.rk3d import RK3DDistributed
        d = RK3DDistributed.__new__(RK3DDistributed)
        d.rank, d.world, d.group, d.slab, d.stream, d._torch = rank, world, None, _FakeSlab(rank, fail), _FakeStream(), _FakeTorch
        d.transport_note, d.transport_log = "", []
        real = dist.get_backend
        dist.get_backend = lambda group=None: backend_name          # 'nccl': auto may go on to rccl
        err = None
        try:
            d._connect(want)
        except RuntimeError as e:
            err = str(e)
        finally:
            dist.get_backend = real
        q.put((rank, d.slab.transport, d.transport_note, d.slab.calls, err, d.transport_log))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("want,fail,backend,expect", [
    ("auto", {}, "gloo", "ipc"),                                   # everything works: IPC
    ("auto", {"ipc_connect": 1}, "gloo", "callback"),              # one rank cannot map its neighbour: every rank drops IPC
    ("auto", {"mismatch": 0}, "nccl", "rccl"),                     # the probe finds wrong data on one rank: on to RCCL (nccl backend)
    ("auto", {"mismatch": "all", "rccl_connect": 1}, "nccl", "callback"),
    ("auto", {"probe": 1}, "gloo", "callback"),                    # one rank cannot even enqueue the probe
    ("auto", {"ipc_init": 0}, "gloo", "callback"),                 # one rank cannot allocate its landing area
    ("ipc", {"ipc_init": 0}, "gloo", "raises"),                    # a named transport that fails raises on EVERY rank
    ("rccl", {}, "nccl", "rccl"),
    ("auto", {"hang": 1, "hang_kinds": ("ipc",)}, "nccl", "rccl"),  # the probe hangs on one rank over IPC: its watchdog fires, all go on to RCCL
    ("rccl", {"hang": 0}, "nccl", "raises"),                       # the deadline serves the RCCL probe too (advisor, round 4)
    ("rccl", {"sizes": 0}, "nccl", "raises"),                      # ranks that disagree on the face message's size never enter ncclCommInitRank
])
def test_transport_selection_is_agreed_on_by_all_ranks(want, fail, backend, expect):
    """RK3DDistributed._connect with a stand-in slab, two ranks over gloo: whatever happens on one rank, both end on the same
    transport (a rank that kept IPC while its neighbour fell back would wait for messages that never come)"""
    world, port = 2, _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    ps = [ctx.Process(target=_connect_worker, args=(r, world, port, q, want, fail, backend)) for r in range(world)]
    for p in ps:
        p.start()
    got = sorted(q.get(timeout=120) for _ in ps)
    for p in ps:
        p.join(timeout=60)
        assert p.exitcode == 0
    kinds = [g[1].split(" ")[0] for g in got]
    if expect == "raises":
        assert all(g[4] and "could not be connected on every rank" in g[4] for g in got) and kinds == ["callback", "callback"]
    else:
        assert kinds == [expect, expect], got
        assert all(g[4] is None for g in got)
        if expect != "callback":
            assert all(g[2] == "in-library " for g in got)
        if fail:
            assert all("disconnect" in g[3] for g in got)           # the dropped transport was disconnected on both ranks
    # every candidate left its verdict in the log, the last one the transport the ranks ended on (or why there is none)
    for g in got:
        log = g[5]
        assert log and all(set(e) == {"transport", "ok", "why"} for e in log)
        assert [e["ok"] for e in log].count(True) == (0 if expect in ("callback", "raises") else 1)
        if expect not in ("callback", "raises"):
            assert log[-1]["ok"] and log[-1]["transport"] == expect
        assert all("sync(deadline)" in c for c in g[3] if c.startswith("sync"))        # no probe is waited for without a deadline
    if fail.get("sizes") is not None:
        assert all("disagree on the size" in g[4] for g in got) and all("rccl_connect" not in g[3] for g in got)


# ------------------------------------------------------------------------------------------------ the 3-D CSF model's ring of slabs
def test_csf_slab_cuts_and_ghost_planes():
    """rk3dcsf.slab_cuts / _SlabGeometry: the slabs' own planes tile the lattice, every slab carries two images at either end, cut around
    the ends of the lattice (the loop wraps z: the slabs form a ring)"""
    from openlbmpm_amd.rk3dcsf import slab_cuts, _SlabGeometry
    assert slab_cuts(40, 1) == [0, 40] and slab_cuts(12, 3) == [0, 4, 8, 12]
    w = np.ones(64); w[:16] = 5.0                     # weights: the planes of the colour interface cost more
    cuts = slab_cuts(64, 4, w)
    assert cuts[0] == 0 and cuts[-1] == 64 and all(b - a >= 4 for a, b in zip(cuts, cuts[1:])) and cuts[1] < 16
    with pytest.raises(ValueError):
        slab_cuts(11, 3)
    a = np.arange(40 * 2 * 3).reshape(40, 2, 3)
    geo = [_SlabGeometry(40, z0, z1) for z0, z1 in zip([0, 14, 27], [14, 27, 40])]
    assert np.array_equal(np.concatenate([g.own(g.cut(a)) for g in geo]), a)
    assert list(geo[0].planes[:3]) == [38, 39, 0] and list(geo[2].planes[-3:]) == [39, 0, 1] and geo[1].slab == (14, 40)
    whole = _SlabGeometry(40, 0, 40)
    assert whole.ghost == (0, 0) and whole.slab is None and np.array_equal(whole.cut(a), a)


class _HostSlab:
    """A host stand-in with the library's slab calls (stage, face_pack / _unpack, face_doubles) and the CSF step's dependency pattern on a
    periodic column of planes: the phase field reads the state one plane around, the 'normal' reads phi two planes around, the new state
    reads n one plane around -- so the same three messages are needed, of two, one and one plane"""
    on_host = True
    W = 3

    def __init__(self, a, params, device=0, diagnostics=False, slab=None):
        self.params, self.slab = params, slab
        self.nz = a.shape[0]
        self.a = np.zeros((self.nz, self.W)); self.phi = np.zeros_like(self.a); self.n = np.zeros_like(self.a)
        self.next, self.steps_done = 0, 0

    def set_macro(self, a, *_):
        self.a[:] = a

    def stage(self, k):
        assert k == self.next
        own = slice(2, self.nz - 2)
        z = np.arange(2, self.nz - 2)
        if k == 0:
            self.phi[own] = 0.25 * self.a[z - 1] + 0.5 * self.a[z] + 0.25 * self.a[z + 1]
        elif k == 1:
            self.n[own] = self.phi[z + 1] - self.phi[z - 1] + 0.1 * (self.phi[z + 2] - self.phi[z - 2])
        else:
            self.a[own] = self.phi[z] + 0.3 * (self.n[z + 1] - self.n[z - 1])
            self.steps_done += 1
        self.next = (k + 1) % 3

    def _planes(self, msg, face, incoming):
        arr = (self.a, self.phi, self.n)[msg]
        k = 2 if msg == 1 else 1
        if incoming:
            return arr, (slice(2 - k, 2) if face == 0 else slice(self.nz - 2, self.nz - 2 + k))
        return arr, (slice(2, 2 + k) if face == 0 else slice(self.nz - 2 - k, self.nz - 2))

    def face_doubles(self, msg, face):
        return (2 if msg == 1 else 1) * self.W

    face_doubles_in = face_doubles

    def _view(self, ptr, n):
        import ctypes
        return np.ctypeslib.as_array((ctypes.c_double * n).from_address(ptr))

    def face_pack(self, msg, face, ptr):
        arr, sl = self._planes(msg, face, False)
        self._view(ptr, arr[sl].size)[:] = arr[sl].reshape(-1)

    def face_unpack(self, msg, face, ptr):
        arr, sl = self._planes(msg, face, True)
        arr[sl] = self._view(ptr, arr[sl].size).reshape(arr[sl].shape)

    def sync(self):
        pass

    def get(self, name):
        return self.a.copy()

    def close(self):
        pass


def _csf_ring_worker(rank, world, port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from openlbmpm_amd.rk3dcsf import RK3DCSFDistributed
        nz = 29
        full = np.random.default_rng(5).standard_normal((nz, _HostSlab.W))
        d = RK3DCSFDistributed(full, dict(tag="host stand-in"), slab_factory=_HostSlab)
        d.set_macro(full, full)
        d.step(7)
        ref = full.copy()
        for _ in range(7):
            phi = 0.25 * np.roll(ref, 1, 0) + 0.5 * ref + 0.25 * np.roll(ref, -1, 0)
            n = np.roll(phi, -1, 0) - np.roll(phi, 1, 0) + 0.1 * (np.roll(phi, -2, 0) - np.roll(phi, 2, 0))
            ref = phi + 0.3 * (np.roll(n, -1, 0) - np.roll(n, 1, 0))
        mine = d.get("a")
        ok = np.array_equal(mine, ref[d.z0:d.z0 + d.nzl]) and d.slab.steps_done == 7
        whole = d.gather(mine)
        ok = ok and ((whole is None) if rank else np.array_equal(whole, ref))
        q.put((rank, bool(ok)))
    except Exception as e:                           # (the parent does not wait for its time-out)
        q.put((rank, repr(e)))
        raise
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 3, 5])
def test_csf_ring_of_slabs_gloo(world):
    """rk3dcsf.RK3DCSFDistributed's orchestration -- three stages per step, a face message after each, the ranks a ring (with two ranks both
    faces join the same pair: told apart by tags) -- with a host stand-in for the solver: equal to the undivided periodic column"""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_csf_ring_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    results = [q.get(timeout=120) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert sorted(results) == [(r, True) for r in range(world)]
