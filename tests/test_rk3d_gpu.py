"""GPU tests of the D3Q19 colour-gradient solver (C ABI): against the independent CPU statement
oracle/rk3d_oracle.c (pinned to the reference by reduction, tests/test_rk3d_reduction.py), and the
slab-decomposed run (k virtual ranks, halo buffers moved exactly as the RCCL path moves them)
against the single-domain run, bit for bit."""
import numpy as np
import pytest

from helpers import rel_err

pytestmark = pytest.mark.gpu
TOL = 1e-10


def _case(nx=40, ny=21, nz=38, seed=4):
    from openlbmpm_amd.geometry import porous_spheres, initial_densities_rk3d
    dom = porous_spheres(nx, ny, nz, porosity=0.7, rmin=3.0, rmax=7.0, seed=seed, nbuf=5)
    rR, rB = initial_densities_rk3d(dom, 5)
    return dom, rR, rB


@pytest.mark.parametrize("layout", ["q23", "dense"])
@pytest.mark.parametrize("relax", ["SRT", "MRT"])
def test_single_slab_vs_oracle(relax, layout, monkeypatch):
    """nx = 40: the 23-value compact storage on row segments shorter than 64 cells (rk3dq_fused<.., RAGGED>), and the dense layout"""
    from openlbmpm_amd.rk3d import RK3DCluster
    from oracle.rk3d import RK3DOracle
    dom, rR, rB = _case()
    par = dict(tauR=1.0, tauB=0.8, relax=relax)
    if layout == "dense":
        monkeypatch.setenv("LBMPM_RK3D_LAYOUT", "dense")
    c = RK3DCluster(dom, 1, par)
    assert c.slabs[0].dominant_kernel == ("rk3dq_fused" if layout == "q23" else "rk3d_fused")
    c.set_density(rR, rB)
    o = RK3DOracle(dom, rR, rB, par)
    for n in (1, 19):
        c.step(n); o.run(n)
        c.observe(); o.macro()
        umax = max(float(np.max(np.abs(o.field(f)))) for f in ("vx", "vy", "vz"))
        for f in ("rhoR", "rhoB", "phi", "vx", "vy", "vz"):
            e = rel_err(c.get(f), o.field(f), scale=umax if f[0] == "v" else None)
            assert e < TOL, "field %s rel err %.3e after %d steps" % (f, e, c.slabs[0].steps_done)
    c.close()


@pytest.mark.parametrize("layout", ["q23", "dense"])
@pytest.mark.parametrize("k", [2, 3, 5])
def test_slabs_equal_single_domain_bitwise(k, layout, monkeypatch):
    from openlbmpm_amd.rk3d import RK3DCluster
    dom, rR, rB = _case(nx=33, ny=18, nz=41, seed=9)
    if layout == "dense":
        monkeypatch.setenv("LBMPM_RK3D_LAYOUT", "dense")
    out = []
    for kk in (1, k):
        c = RK3DCluster(dom, kk)
        c.set_density(rR, rB)
        c.step(15)
        c.observe()
        out.append({f: c.get(f) for f in ("rhoR", "rhoB", "phi", "vx", "vy", "vz")})
        c.close()
    for f in out[0]:
        assert np.array_equal(out[0][f], out[1][f]), f


@pytest.mark.parametrize("env", [{"LBMPM_RK3D_BOUNDARY": "4"}, {"LBMPM_RK3D_VARIANT": "1"}, {"LBMPM_RK3D_TILE": "1"},
                                 {"LBMPM_RK3D_TILE": "2", "LBMPM_RK3D_CHUNK": "5"}, {"LBMPM_RK3D_FILL": "0"}],
                         ids=lambda e: ",".join("%s=%s" % (k[11:], v) for k, v in e.items()))
def test_kernel_schedules_agree_bitwise(env, monkeypatch, knobs):
    """split sweeps, other tile shapes / chunk lengths, the interior|boundary split used to overlap
    the halo exchange: all the same arithmetic, so the same bits as the default single-slab run"""
    from openlbmpm_amd.rk3d import RK3DCluster
    dom, rR, rB = _case(nx=70, ny=19, nz=41, seed=11)
    monkeypatch.setenv("LBMPM_RK3D_LAYOUT", "dense")          # (the default for any nx is the 23-value compact storage: other rounding)
    ref = RK3DCluster(dom, 1)
    assert ref.slabs[0].dominant_kernel == "rk3d_fused"
    ref.set_density(rR, rB)
    ref.step(9); ref.observe()
    knobs(env)
    c = RK3DCluster(dom, 3)
    c.set_density(rR, rB)
    c.step(9); c.observe()
    for f in ("rhoR", "rhoB", "phi", "vx", "vy", "vz"):
        assert np.array_equal(ref.get(f), c.get(f)), f
    ref.close(); c.close()


@pytest.mark.parametrize("storage", ["23", "38"])
@pytest.mark.parametrize("relax", ["SRT", "MRT"])
def test_compact_storage_vs_oracle(relax, storage, monkeypatch):
    """nx a multiple of 64 selects the compact (fluid-cells-only) storage: by default 23 doubles per cell (rk3dq_fused: 19
    colour-blind populations + k_R + the recolouring vector), with LBMPM_RK3D_STORAGE=38 both colour lattices (rk3dc_fused)"""
    from openlbmpm_amd.rk3d import RK3DCluster
    from oracle.rk3d import RK3DOracle
    dom, rR, rB = _case(nx=128, ny=21, nz=30, seed=5)
    par = dict(tauR=1.0, tauB=0.8, relax=relax)
    monkeypatch.setenv("LBMPM_RK3D_STORAGE", storage)
    c = RK3DCluster(dom, 1, par)
    assert c.slabs[0].dominant_kernel == ("rk3dq_fused" if storage == "23" else "rk3dc_fused")
    c.set_density(rR, rB)
    o = RK3DOracle(dom, rR, rB, par)
    for n in (1, 14):
        c.step(n); o.run(n)
        c.observe(); o.macro()
        umax = max(float(np.max(np.abs(o.field(f)))) for f in ("vx", "vy", "vz"))
        for f in ("rhoR", "rhoB", "phi", "vx", "vy", "vz"):
            e = rel_err(c.get(f), o.field(f), scale=umax if f[0] == "v" else None)
            assert e < TOL, "field %s rel err %.3e after %d steps" % (f, e, c.slabs[0].steps_done)
    c.close()


@pytest.mark.parametrize("env,k", [({}, 2), ({}, 3), ({"LBMPM_RK3D_BOUNDARY": "4"}, 3), ({"LBMPM_RK3D_TILE": "1"}, 1),
                                   ({"LBMPM_RK3D_CHUNK": "7"}, 2)],
                         ids=lambda e: ",".join("%s=%s" % (k[11:], v) for k, v in e.items()) if isinstance(e, dict) else "k%d" % e)
def test_compact_storage_equals_dense_bitwise(env, k, monkeypatch, knobs):
    """the 38-value compact storage is the dense layout with the solid cells left out: the same bits"""
    from openlbmpm_amd.rk3d import RK3DCluster
    dom, rR, rB = _case(nx=64, ny=19, nz=41, seed=12)
    monkeypatch.setenv("LBMPM_RK3D_STORAGE", "38")
    monkeypatch.setenv("LBMPM_RK3D_LAYOUT", "dense")
    ref = RK3DCluster(dom, 1)
    assert ref.slabs[0].dominant_kernel == "rk3d_fused"
    ref.set_density(rR, rB)
    ref.step(11); ref.observe()
    monkeypatch.delenv("LBMPM_RK3D_LAYOUT")
    knobs(env)
    c = RK3DCluster(dom, k)
    assert c.slabs[0].dominant_kernel == "rk3dc_fused"
    c.set_density(rR, rB)
    c.step(11); c.observe()
    for f in ("rhoR", "rhoB", "phi", "vx", "vy", "vz"):
        assert np.array_equal(ref.get(f), c.get(f)), f
    ref.close(); c.close()


@pytest.mark.parametrize("env,k", [({"LBMPM_RK3D_LAYOUT": "dense"}, 1), ({"LBMPM_RK3D_LAYOUT": "dense", "LBMPM_RK3D_VARIANT": "1"}, 2),
                                   ({}, 3), ({"LBMPM_RK3D_TILE": "1", "LBMPM_RK3D_CHUNK": "6"}, 2)],
                         ids=lambda e: ",".join("%s=%s" % (k[11:], v) for k, v in e.items()) if isinstance(e, dict) else "k%d" % e)
def test_mrt_schedules_agree_bitwise(env, k, monkeypatch, knobs):
    """MRT relaxation: dense / compact (38-value) storage, split sweeps, slab decomposition -- one arithmetic"""
    from openlbmpm_amd.rk3d import RK3DCluster
    dom, rR, rB = _case(nx=64, ny=19, nz=41, seed=12)
    par = dict(relax="MRT", tauR=0.9, tauB=0.7)
    monkeypatch.setenv("LBMPM_RK3D_STORAGE", "38")
    ref = RK3DCluster(dom, 1, par)
    ref.set_density(rR, rB)
    ref.step(11); ref.observe()
    knobs(env)
    c = RK3DCluster(dom, k, par)
    c.set_density(rR, rB)
    c.step(11); c.observe()
    for f in ("rhoR", "rhoB", "phi", "vx", "vy", "vz"):
        assert np.array_equal(ref.get(f), c.get(f)), f
    ref.close(); c.close()


@pytest.mark.parametrize("relax", ["SRT", "MRT"])
def test_q23_storage_equals_the_38_value_kernels_to_roundoff(relax, monkeypatch):
    """rk3dq_fused stores 19 colour-blind populations + k_R + A per cell and rebuilds the colours while pulling: the same step as
    rk3dc_fused, different rounding.  Both colours present in every cell (no single-colour shortcut anywhere), 30 steps."""
    from openlbmpm_amd.rk3d import RK3DCluster
    dom, _, _ = _case(nx=128, ny=24, nz=40, seed=21)
    rng = np.random.default_rng(3)
    w = 0.2 + 0.6 * rng.random(dom.shape)
    rR = np.where(dom == 1, w, 0.0); rB = np.where(dom == 1, 1.0 - w, 0.0)
    par = dict(relax=relax, tauR=0.9, tauB=0.75)
    out = []
    for storage in ("23", "38"):
        monkeypatch.setenv("LBMPM_RK3D_STORAGE", storage)
        c = RK3DCluster(dom, 1, par)
        assert c.slabs[0].dominant_kernel == ("rk3dq_fused" if storage == "23" else "rk3dc_fused")
        c.set_density(rR, rB)
        c.step(30); c.observe()
        out.append({f: c.get(f) for f in ("rhoR", "rhoB", "phi", "vx", "vy", "vz")})
        c.close()
    umax = max(float(np.max(np.abs(out[1][f]))) for f in ("vx", "vy", "vz"))
    for f in out[0]:
        assert rel_err(out[0][f], out[1][f], scale=umax if f[0] == "v" else None) < 1e-11, f


@pytest.mark.parametrize("env,k", [({}, 2), ({}, 3), ({}, 5), ({"LBMPM_RK3D_BOUNDARY": "4"}, 3), ({"LBMPM_RK3D_CHUNK": "7"}, 2)],
                         ids=lambda e: ",".join("%s=%s" % (k[11:], v) for k, v in e.items()) if isinstance(e, dict) else "k%d" % e)
@pytest.mark.parametrize("relax", ["SRT", "MRT"])
def test_q23_slabs_equal_single_domain_bitwise(relax, env, k, monkeypatch, knobs):
    """q23 storage, k virtual ranks: ONE face message per cut and step (populations, records, row flags and the class sums from which
    the neighbour completes the phase field of its halo plane) -- the same bits as the single domain, mixed and single-colour
    regions alike (the initial state is red below a blue buffer: most row segments carry a flag instead of records)"""
    from openlbmpm_amd.rk3d import RK3DCluster
    dom, rR, rB = _case(nx=128, ny=21, nz=38, seed=5)
    par = dict(relax=relax, tauR=0.9, tauB=0.7)
    ref = RK3DCluster(dom, 1, par)
    assert ref.slabs[0].dominant_kernel == "rk3dq_fused"
    ref.set_density(rR, rB)
    knobs(env)
    c = RK3DCluster(dom, k, par)
    assert c.slabs[0].dominant_kernel == "rk3dq_fused" and c.slabs[0].one_exchange
    c.set_density(rR, rB)
    for n in (1, 12):
        ref.step(n); ref.observe(); c.step(n); c.observe()
        for f in ("rhoR", "rhoB", "phi", "vx", "vy", "vz"):
            assert np.array_equal(ref.get(f), c.get(f)), (f, n)
    ref.close(); c.close()


def test_q23_slabs_of_a_wide_lattice_equal_single_domain_bitwise():
    """The bench's lattice shape in small: 192 x 192 planes = three row segments and 1152 waves in the face kernels when a rank has a
    neighbour on both sides.  Round 3 lost whole waves' worth of face-message entries exactly there (the face kernels then kept a
    private copy of their argument struct in scratch memory; tests/test_codeobj.py keeps scratch out of these kernels): every face
    message must come out the same every time, and three slabs must equal the single domain bit for bit, phase by phase
    (RK3DCluster) and through the pipelined lbmpm_rk3d_step_slab with one host thread per rank."""
    import threading
    import torch
    import bench
    from openlbmpm_amd.rk3d import RK3DCluster, RK3DDistributed, RK3DSlab
    n, K, steps = 192, 3, 5
    dom = bench.c5_domain((n, n, n))
    rR, rB = bench.c5_densities(dom, 0, n)
    par = dict(relax="MRT")
    ref = RK3DSlab(dom, 0, n, par); ref.set_density(rR, rB); ref.step_single(steps); ref.phase_field(diagnostics=True)
    want = {f: ref.get(f) for f in ("phi", "rhoR", "vz")}
    ref.close()
    assert np.isfinite(want["phi"]).all()
    c = RK3DCluster(dom, K, par); c.set_density(rR, rB)
    mid = c.slabs[1]
    for name in ("f_send_up", "f_send_down"):            # the rank with two neighbours packs both faces in one launch
        t, first = mid.buffer(name), None
        for _ in range(5):
            with torch.cuda.stream(c.stream):
                t.zero_(); mid.pack()
            c.stream.synchronize()
            img = t.clone()
            first = img if first is None else first
            assert torch.equal(img, first), name
    c.step(steps); c.observe(); c.stream.synchronize()
    for f in want:
        assert np.array_equal(c.get(f), want[f]), ("cluster", f)
    c.close()
    # the pipelined step as the ranks of a real run make it, with an exact exchange: a barrier inside the callback
    parts = RK3DDistributed.partition(dom, K)
    st = torch.cuda.Stream(0)
    slabs = []
    for z0, nz in parts:
        s = RK3DSlab(dom, z0, nz, par); s.set_density(rR[z0:z0 + nz], rB[z0:z0 + nz]); s.use_torch_stream(st); slabs.append(s)
    gate, errors = threading.Barrier(K), []

    def rank(r):
        s, below, above = slabs[r], r > 0, r + 1 < K

        def exchange(what):
            gate.wait()
            with torch.cuda.stream(st):
                if below: s.buffer("f_recv_below").copy_(slabs[r - 1].buffer("f_send_up"))
                if above: s.buffer("f_recv_above").copy_(slabs[r + 1].buffer("f_send_down"))
            gate.wait()
        try:
            s.step_slab(steps, below, above, exchange)
        except BaseException as e:      # noqa: BLE001 -- reported by the main thread
            errors.append(e); gate.abort()
    th = [threading.Thread(target=rank, args=(r,)) for r in range(K)]
    for t in th: t.start()
    for t in th: t.join()
    assert not errors, errors
    with torch.cuda.stream(st):
        for s in slabs: s.pack()
        for r, s in enumerate(slabs):
            if r > 0: s.buffer("f_recv_below").copy_(slabs[r - 1].buffer("f_send_up"))
            if r + 1 < K: s.buffer("f_recv_above").copy_(slabs[r + 1].buffer("f_send_down"))
        for r, s in enumerate(slabs):
            s.unpack(r > 0, r + 1 < K); s.phase_field(diagnostics=True)
    for f in want:
        assert np.array_equal(np.concatenate([s.get(f) for s in slabs], axis=0), want[f]), ("pipelined", f)
    for s in slabs: s.close()


def test_mrt_differs_from_srt_and_unknown_relaxation_is_rejected():
    from openlbmpm_amd.rk3d import RK3DCluster
    dom, rR, rB = _case(nx=24, ny=12, nz=20, seed=2)
    out = []
    for relax in ("SRT", "MRT"):
        c = RK3DCluster(dom, 1, dict(relax=relax))
        c.set_density(rR, rB)
        c.step(6); c.observe()
        out.append(c.get("vz"))
        c.close()
    assert np.abs(out[0] - out[1]).max() > 1e-9
    with pytest.raises(ValueError):
        RK3DCluster(dom, 1, dict(relax="TRT"))


def test_single_slab_convenience_equals_phases():
    from openlbmpm_amd.rk3d import RK3DCluster, RK3DSlab
    dom, rR, rB = _case(nx=24, ny=12, nz=20, seed=2)
    s = RK3DSlab(dom, 0, dom.shape[0])
    s.set_density(rR, rB)
    s.step_single(7)
    s.phase_field(diagnostics=True)
    c = RK3DCluster(dom, 1)
    c.set_density(rR, rB)
    c.step(7); c.observe()
    assert np.array_equal(s.get("phi"), c.get("phi")) and np.array_equal(s.get("vz"), c.get("vz"))
    s.close(); c.close()


def _free_port():
    import socket
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        return sk.getsockname()[1]


def _two_rank_run(tmp_path, backend, same_gpu, nx=33, transport="callback", extra_env=None):
    """two OS processes through RK3DDistributed (one lbmpm_rk3d_step_slab call for all steps, the exchanges as
    callbacks); returns the gathered fields and the per-rank timing dicts"""
    import json
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    script = tmp_path / "w.py"
    script.write_text('''
import json, os, sys
sys.path.insert(0, %r); sys.path.insert(0, os.path.join(%r, "tests"))
import numpy as np, torch, torch.distributed as dist
from test_rk3d_gpu import _case
from openlbmpm_amd.rk3d import RK3DDistributed
rank = int(os.environ["RANK"])
dev = 0 if %r else rank
torch.cuda.set_device(dev)
if %r == "nccl":
    dist.init_process_group("nccl", device_id=torch.device("cuda", dev))
else:
    dist.init_process_group("gloo")
dom, rR, rB = _case(nx=int(os.environ["LBMPM_TEST_NX"]), ny=18, nz=41, seed=9)
d = RK3DDistributed(dom, json.loads(os.environ.get("LBMPM_TEST_PAR", "{}")) or None, device=dev, transport=os.environ["LBMPM_TEST_TRANSPORT"])
d.set_density(rR, rB)
d.step(9); d.step(6, timed=True)
t = d.timing()
d.observe()
np.save(os.path.join(%r, "phi_%%d.npy" %% rank), d.slab.get("phi"))
np.save(os.path.join(%r, "vz_%%d.npy" %% rank), d.slab.get("vz"))
json.dump(dict(t, backend=dist.get_backend(), world=dist.get_world_size()), open(os.path.join(%r, "t_%%d.json" %% rank), "w"))
d.close(); dist.destroy_process_group()
''' % (root, root, same_gpu, backend, str(tmp_path), str(tmp_path), str(tmp_path)))
    subprocess.check_call([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
                           "--master-addr", "127.0.0.1", "--master-port", str(_free_port()), str(script)],
                          env=dict(os.environ, LBMPM_TEST_NX=str(nx), LBMPM_TEST_TRANSPORT=transport, **(extra_env or {})), timeout=300)
    fields = {f: np.concatenate([np.load(tmp_path / ("%s_%d.npy" % (f, r))) for r in range(2)], axis=0) for f in ("phi", "vz")}
    return fields, [json.load(open(tmp_path / ("t_%d.json" % r))) for r in range(2)]


def _single_process_reference(nx=33, par=None):
    from openlbmpm_amd.rk3d import RK3DCluster
    dom, rR, rB = _case(nx=nx, ny=18, nz=41, seed=9)
    c = RK3DCluster(dom, 1, par)
    c.set_density(rR, rB)
    c.step(15); c.observe()
    ref = {f: c.get(f) for f in ("phi", "vz")}
    c.close()
    return ref


@pytest.mark.parametrize("nx,transport,env", [(33, "callback", {"LBMPM_RK3D_LAYOUT": "dense"}), (64, "callback", None), (64, "ipc", None),
                                              (64, "ipc", {"LBMPM_IPC_FLAG_KERNELS": "1"}), (96, "ipc", None), (33, "callback", None),
                                              (64, "ipc", {"LBMPM_TEST_PAR": '{"outlet": "Convective", "inlet": "Dirichlet", "densityRH": 1e-8, "densityBH": 1.003, "relax": "MRT"}'})],
                         ids=["dense-two-exchanges-callback", "q23-one-exchange-callback", "q23-ipc-stream-value-ops", "q23-ipc-flag-kernels",
                              "q23-nx96-ipc", "q23-nx33-callback", "q23-ipc-convective-outlet-pressure-inlet"])
def test_two_process_slab_run_equals_single_process(tmp_path, nx, transport, env, monkeypatch):
    """The N>1 orchestration of bench.py / RK3DDistributed with two OS processes sharing this one GPU: the gathered result must
    equal the single-process run bit for bit; the per-phase timing is filled in.
    callback: the exchange as a Python callback per step over the gloo transport staged through the host (RCCL refuses duplicate
    devices).  nx = 64: the bench's storage -- boundary planes first, ONE face exchange beside the interior planes; nx = 96, 33: the
    same on row segments of 48 and 33 cells (any nx runs the 23-value storage; LBMPM_RK3D_LAYOUT=dense keeps the two-exchange layout).
    ipc: the transport INSIDE the library (include/lbmpm.h LBMPM_TRANSPORT_IPC) -- each process maps the other's landing area with
    hipIpcOpenMemHandle, a message is one hipMemcpyAsync + hipStreamWriteValue64, the receiver's stream waits with
    hipStreamWaitValue64; no callback, torch.distributed only carries the handles at set-up.  flag-kernels: the same with the
    one-lane kernels that devices without stream value operations fall back on."""
    got, timing = _two_rank_run(tmp_path, "gloo", same_gpu=True, nx=nx, transport=transport, extra_env=env)
    for k, v in (env or {}).items():
        if k.startswith("LBMPM_RK3D"):
            monkeypatch.setenv(k, v)
    import json
    ref = _single_process_reference(nx, json.loads((env or {}).get("LBMPM_TEST_PAR", "null")))
    for f in ref:
        assert np.array_equal(got[f], ref[f]), f
    for t in timing:
        assert t["world"] == 2 and t["steps"] == 6 and t["step_ms"] > 0 and t["interior_ms"] > 0 and t["boundary_ms"] > 0
        assert t["bytes_per_face"] > 0 and t["step_ms"] >= t["boundary_ms"]
        if transport == "ipc":
            assert "in-library ipc" in t["transport"] and ("one-lane flag kernels" if env and "LBMPM_IPC_FLAG_KERNELS" in env else "stream value operations") in t["transport"]
        else:
            assert t["transport"] == "callback"


def test_a_rank_whose_neighbour_dies_mid_run_raises_at_its_deadline(tmp_path):
    """The steady-state watchdog of the IPC transport (lbmpm_rk3d_sync_deadline; round 4 had a deadline at set-up only: a rank that
    died mid-run left its neighbour's stream in hipStreamWaitValue64 for good).  Two OS processes on this GPU over the in-library IPC
    transport; rank 1 leaves without a word (os._exit) after 5 steps, rank 0 goes on stepping: its stream blocks in the wait for a
    message that never comes, sync() gives up after its 3 s deadline with status LBMPM_ERR_TIMEOUT, the released context refuses
    further exchanges with the same status and can be closed.  The same with the one-lane flag kernels."""
    import json
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    script = tmp_path / "k.py"
    script.write_text('''
import json, os, sys, time
sys.path.insert(0, %r); sys.path.insert(0, os.path.join(%r, "tests"))
import numpy as np, torch, torch.distributed as dist
from test_rk3d_gpu import _case
from openlbmpm_amd import _lib
from openlbmpm_amd.rk3d import RK3DDistributed
rank = int(os.environ["RANK"])
torch.cuda.set_device(0)
dist.init_process_group("gloo")
dom, rR, rB = _case(nx=64, ny=18, nz=41, seed=9)
d = RK3DDistributed(dom, device=0, transport="ipc")
d.deadline_s = 3.0
d.set_density(rR, rB)
d.step(5); d.sync()
if rank == 1:
    os._exit(0)                      # dies without closing anything
out = dict(transport=d.transport)
t0 = time.perf_counter()
try:
    d.step(40)                       # enqueues 40 steps: the second one waits for rank 1's message
    d.sync()
    out["raised"] = None
except _lib.LbmpmError as e:
    out["raised"], out["status"], out["seconds"] = str(e), e.status, time.perf_counter() - t0
try:
    d.step(1)
    out["after"] = None
except _lib.LbmpmError as e:
    out["after"] = e.status
d.close()
json.dump(out, open(os.path.join(%r, "rank0.json"), "w"))
os._exit(0)                          # (no destroy_process_group: the peer is gone)
''' % (root, root, str(tmp_path)))
    for env in (None, {"LBMPM_IPC_FLAG_KERNELS": "1"}):
        if os.path.exists(tmp_path / "rank0.json"):
            os.remove(tmp_path / "rank0.json")
        subprocess.call([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
                         "--master-port", str(_free_port()), str(script)], env=dict(os.environ, **(env or {})), timeout=240)
        out = json.load(open(tmp_path / "rank0.json"))
        assert out["transport"].startswith("ipc"), out
        assert out["raised"] and out["status"] == -6 and "did not arrive" in out["raised"], out
        assert 2.5 < out["seconds"] < 30., out
        assert out["after"] == -6, out


def test_a_named_transport_on_the_two_exchange_storage_raises(tmp_path):
    """advisor, round 4: transport='ipc' on a lattice that runs the 38-value storage (here: LBMPM_RK3D_LAYOUT=dense; two exchanges per step)
    was silently served by the callback; a named transport that cannot be had raises (on every rank alike: the decision depends on
    the lattice only), and 'auto' says in its log why it fell back"""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    script = tmp_path / "n.py"
    script.write_text('''
import os, sys
sys.path.insert(0, %r); sys.path.insert(0, os.path.join(%r, "tests"))
import torch, torch.distributed as dist
from test_rk3d_gpu import _case
from openlbmpm_amd.rk3d import RK3DDistributed
torch.cuda.set_device(0)
dist.init_process_group("gloo")
dom, rR, rB = _case(nx=33, ny=18, nz=41, seed=9)
try:
    RK3DDistributed(dom, device=0, transport="ipc")
    raise SystemExit("no error")
except RuntimeError as e:
    assert "two exchanges per step" in str(e), e
d = RK3DDistributed(dom, device=0, transport="auto")
assert d.transport == "callback" and d.transport_log and not d.transport_log[0]["ok"] and "two exchanges" in d.transport_log[0]["why"]
d.set_density(rR, rB); d.step(2); d.sync(); d.close()
dist.destroy_process_group()
''' % (root, root))
    subprocess.check_call([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
                           "--master-port", str(_free_port()), str(script)], env=dict(os.environ, LBMPM_RK3D_LAYOUT="dense"), timeout=240)


@pytest.mark.parametrize("kind", ["ipc", "rccl"])
def test_transport_selftest_on_one_gpu(kind):
    """The transports of the slab exchange at transport level, on the one GPU of this box (lbmpm_transport_selftest): three messages
    up and down to the caller itself.  ipc: the landing area connected to itself, hipMemcpyAsync + stream value operations, both slot
    parities.  rccl: librccl opened with dlopen, a one-rank communicator (ncclCommInitRank), ncclSend / ncclRecv to rank 0 in one
    group on a stream -- the RCCL branch of the library executing on hardware (with two ranks it needs two GPUs: the test below)."""
    from openlbmpm_amd import _lib
    from openlbmpm_amd.rk3d import _torch_librccl
    import torch  # noqa: F401  (its HIP runtime first, as everywhere)
    L = _lib.lib()
    path = _torch_librccl()
    _lib.check(L.lbmpm_transport_selftest(_lib.TRANSPORT_IPC if kind == "ipc" else _lib.TRANSPORT_RCCL, 0, 3 * 1024 * 1024 + 8,
                                          path.encode() if path else None), "lbmpm_transport_selftest(%s)" % kind)


def test_a_transport_needs_the_compact_storage_and_matching_neighbours(monkeypatch):
    """error paths of the connect calls: the dense storage has no in-library transport; a blob for a neighbour the slab lacks, a
    blob that is none and a neighbour that expects another message size are refused with a status, nothing hangs"""
    from openlbmpm_amd import _lib
    from openlbmpm_amd._lib import LbmpmError
    from openlbmpm_amd.rk3d import RK3DSlab
    dom, rR, rB = _case(nx=64, ny=18, nz=41, seed=9)
    lo, hi = RK3DSlab(dom, 0, 20), RK3DSlab(dom, 20, 21)
    b_lo, b_hi = lo.ipc_init(), hi.ipc_init()
    with pytest.raises(LbmpmError) as e:
        lo.ipc_connect(b_hi, None)                 # the bottom slab has no rank below
    assert e.value.status == _lib.ERR_INVALID
    with pytest.raises(LbmpmError) as e:
        lo.ipc_connect(None, b"\0" * 256)
    assert e.value.status == _lib.ERR_INVALID and "not a blob" in str(e.value)
    other = RK3DSlab(dom, 21, 20)                  # another cut: its bottom plane is not the plane above `lo`
    with pytest.raises(LbmpmError) as e:
        lo.ipc_connect(None, other.ipc_init())
    assert e.value.status == _lib.ERR_INVALID and "different cuts" in str(e.value)
    lo.ipc_connect(None, b_hi); hi.ipc_connect(b_lo, None)          # same process: connected by pointer
    assert lo.transport.startswith("ipc") and hi.transport.startswith("ipc")
    lo.transport_disconnect()
    assert lo.transport == "callback"
    dense_dom, _, _ = _case(nx=33, ny=18, nz=41, seed=9)
    monkeypatch.setenv("LBMPM_RK3D_LAYOUT", "dense")
    d = RK3DSlab(dense_dom, 0, 20)
    with pytest.raises(LbmpmError) as e:
        d.ipc_init()
    assert e.value.status == _lib.ERR_INVALID
    for s in (lo, hi, other, d):
        s.close()


def test_two_gpu_rccl_slab_run_equals_single_process(tmp_path):
    """the same over the real transport: backend nccl = RCCL, one GPU per rank (needs a box with two GPUs)"""
    import torch
    if torch.cuda.device_count() < 2:
        pytest.skip("RCCL refuses two ranks on one device: needs >= 2 GPUs")
    for transport in ("rccl", "ipc", "callback"):
        d = tmp_path / transport
        d.mkdir()
        got, timing = _two_rank_run(d, "nccl", same_gpu=False, nx=64, transport=transport)
        ref = _single_process_reference(64)
        for f in ref:
            assert np.array_equal(got[f], ref[f]), (transport, f)
        assert all(t["backend"] == "nccl" and t["world"] == 2 and transport in t["transport"] for t in timing)


def test_bench_line_of_a_two_rank_run(tmp_path):
    """`bench.py --gpus 2` end to end (two ranks sharing this GPU over the gloo rehearsal transport): one JSON line from rank
    0 with the whole-job value and, per rank, what the slab step measured -- the fields the N > 1 record is read by"""
    import json
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    out = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
                          "--master-port", str(_free_port()), os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "6", "--warmup", "2",
                          "--size", "128", "128", "128"], env=dict(os.environ, LBMPM_DIST_BACKEND="gloo"), capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [ln for ln in out.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["steps"] == 6 and d["scaling"] == "strong" and d["value"] > 0 and d["unit"] == "MLUPS"
    m = d["multi_gpu"]
    # two processes on one GPU: the in-library IPC transport connects (auto), gloo carries the handles
    assert m["backend"] == "gloo" and m["world_size"] == 2 and m["boundary_depth_planes"] == 2 and "in-library ipc" in m["transport"]
    assert len(m["host_enqueue_us_per_step"]) == 2 and all(v > 0 for v in m["host_enqueue_us_per_step"])
    assert [r["rank"] for r in m["per_rank"]] == [0, 1]
    for r in m["per_rank"]:
        assert r["steps"] == 6 and r["step_ms"] > 0 and r["interior_ms"] > 0 and r["boundary_ms"] > 0 and r["bytes_per_face"] > 0
        assert r["schedule"].startswith("boundary planes -> one face exchange")       # 128^3: the q23 storage
        assert 0.0 <= r["exchange_exposed_ms"] <= r["step_ms"]
    planes = [r["planes"] for r in m["per_rank"]]
    assert planes[0][0] == 0 and planes[0][1] == planes[1][0] and planes[1][1] == 128
    assert sum(r["fluid_nodes"] for r in m["per_rank"]) == d["config"]["fluid_nodes"]
    # round 5: which transport every rank ended on, what set-up tried before it and why a candidate was rejected, the device of each rank
    assert m["transport_per_rank"] == [r["transport"] for r in m["per_rank"]] and all("in-library ipc" in t for t in m["transport_per_rank"])
    tried = m["transport_candidates"][0]["tried"]
    assert m["transport_candidates"][0]["rank"] == 0 and tried[-1]["transport"] == "ipc" and tried[-1]["ok"] and "probe" in tried[-1]["why"]
    assert all(r["device"].startswith("cuda:0") for r in m["per_rank"]) and "re-cut" in m["partition"]


def test_bench_starts_its_own_ranks():
    """`python bench.py --gpus 2` with NO launcher (the form the round-end driver runs): bench.py starts the two ranks itself and
    rank 0 prints one line with n_gpus 2 (gloo rehearsal transport on this one GPU); on the real transport it refuses a node
    with fewer GPUs than ranks instead of quietly running one"""
    import json
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    out = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "5", "--warmup", "1", "--size", "128", "128", "128"],
                         env=dict(env, LBMPM_DIST_BACKEND="gloo"), capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [ln for ln in out.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["steps"] == 5 and d["multi_gpu"]["world_size"] == 2 and d["windows"]["count"] == 5
    import torch
    if torch.cuda.device_count() < 2:
        out = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "2"], env=env, capture_output=True, text=True, timeout=300)
        assert out.returncode != 0 and "one rank per GPU" in (out.stderr + out.stdout)


def test_bench_with_eight_ranks_on_a_wide_lattice():
    """`python bench.py --gpus 8` as the round-end driver runs it on an 8-GPU node, rehearsed here with eight processes sharing this GPU
    (gloo): a 192 x 192 x 128 lattice -- three row segments, six ranks with a neighbour on both sides -- must come through with finite
    fields (the bench checks mass and finiteness at the end of the timed region) and one line from rank 0"""
    import json
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    out = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "8", "--steps", "5", "--warmup", "2", "--size", "192", "192", "128"],
                         env=dict(env, LBMPM_DIST_BACKEND="gloo"), capture_output=True, text=True, timeout=900)
    assert out.returncode == 0, (out.stdout + out.stderr)[-3000:]
    lines = [ln for ln in out.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1
    d = json.loads(lines[0])
    m = d["multi_gpu"]
    assert d["n_gpus"] == 8 and m["world_size"] == 8 and [r["rank"] for r in m["per_rank"]] == list(range(8)) and d["value"] > 0
    planes = [r["planes"] for r in m["per_rank"]]
    assert planes[0][0] == 0 and planes[-1][1] == 128 and all(planes[i][1] == planes[i + 1][0] for i in range(7))
    assert sum(r["fluid_nodes"] for r in m["per_rank"]) == d["config"]["fluid_nodes"]


@pytest.mark.parametrize("medium", ["porous, red-wetting grains", "open duct, neutral walls"])
def test_row_flags_while_a_front_sweeps_the_lattice(monkeypatch, medium, knobs):
    """The row flags of the 23-value storage under a moving interface.  Blue is driven in at 2e-2 lattice units through a porous
    lattice of two row segments per row; over 1600 steps the front crosses more than 40 planes, i.e. row segments go single-colour ->
    mixed -> single-colour of the other colour (the minority colour's tail is exactly zero some 30 planes away from the interface:
    below 1e-16 of the density it does not survive the sums), and with marching chunks of 8 planes (LBMPM_RK3D_CHUNK) it crosses
    chunk borders all the way.  At five
    checkpoints: (a) densities and phase field equal the 38-value kernels' (both colour lattices stored, no flags) within 1e-9;
    (b) the stored flag of every row segment equals "every fluid cell of the segment holds one colour", recomputed on the host from
    the densities the step collided with -- a flagged segment holds one colour only (the other's density is within 2^-51 of the total's
    in every cell: zero, or a rounding residue of rho - rho_R, which the collision treats as absent), an unflagged one holds a cell with both; (c) three slabs (one face message per cut and step, flags travelling with it) give the single domain's bits."""
    from openlbmpm_amd.rk3d import RK3DCluster
    from openlbmpm_amd.geometry import porous_spheres, initial_densities_rk3d
    dom = porous_spheres(128, 24, 192, porosity=0.7, rmin=3.0, rmax=7.0, seed=31, nbuf=5)
    par = dict(relax="MRT", tauR=1.0, tauB=1.0, velocityZB=-2.0e-2)
    if medium.startswith("open"):           # nothing holds the displaced colour back: the segments behind the front that do not touch
        dom = np.ones((192, 24, 256), dtype=np.uint8)      # a side wall (the middle two of four) turn single-colour again
        dom[:, :, 0] = 0; dom[:, :, -1] = 0
        par.update(SolidRhoR=0.5, SolidRhoB=0.5)
    rR, rB = initial_densities_rk3d(dom, 5)
    knobs({"LBMPM_RK3D_CHUNK": "8"})
    runs = {}
    for name, storage, k in (("q23", "23", 1), ("q23 x3", "23", 3), ("both lattices", "38", 1)):
        monkeypatch.setenv("LBMPM_RK3D_STORAGE", storage)
        c = RK3DCluster(dom, k, par)
        assert c.slabs[0].dominant_kernel == ("rk3dq_fused" if storage == "23" else "rk3dc_fused")
        c.set_density(rR, rB)
        runs[name] = c
    nz, ny, nx = dom.shape
    fluid = dom == 1
    seg_has_fluid = fluid.reshape(nz, ny, nx // 64, 64).any(axis=3)
    front_planes = []
    done = 0
    for target in (200, 500, 800, 1200, 1600):
        for c in runs.values():
            c.step(target - 1 - done)
            c.observe()                      # the streamed, boundary-corrected state: the densities step `target` collides with
        obs = {name: {f: c.get(f) for f in ("rhoR", "rhoB", "phi")} for name, c in runs.items()}
        for c in runs.values():
            c.step(1)
        done = target
        a, b = obs["q23"], obs["both lattices"]
        for f in ("rhoR", "rhoB", "phi"):
            assert np.isfinite(a[f]).all()
            assert rel_err(a[f], b[f]) < 1e-9, (target, f, rel_err(a[f], b[f]))
            assert np.array_equal(a[f], obs["q23 x3"][f]), (target, f)
        # (b) flags written by step `target` (planes next to the ghost planes included; the ghost planes themselves are never collided)
        s = runs["q23"].slabs[0]
        # (the collision's rule, collide_store in csrc/rk3d.hip: a colour with |rho_c| <= 2^-51 rho -- absent, or a rounding residue of
        #  rho - rho_R -- is absent)
        tiny = 2.0 ** -51 * (a["rhoR"] + a["rhoB"])
        only_red = np.where(fluid, np.abs(a["rhoB"]) <= tiny, True).reshape(nz, ny, nx // 64, 64).all(axis=3)
        only_blue = np.where(fluid, np.abs(a["rhoR"]) <= tiny, True).reshape(nz, ny, nx // 64, 64).all(axis=3)
        want = only_red.astype(np.int64) + 2 * only_blue.astype(np.int64)
        mixed_planes = 0
        for z in range(1, nz - 1):
            got = s.debug_plane(24, z + 1).astype(np.int64)           # local plane index = z + 1 (plane 0 is the halo plane)
            sel = seg_has_fluid[z]
            assert np.array_equal(got[sel], want[z][sel]), (target, z, got[sel][:8], want[z][sel][:8])
            mixed_planes += int((got[sel] == 0).any())
        # where the front is: the highest plane whose segments are all still pure red (the outlet plane pair at the bottom imposes
        # blue, so the lowest planes are mixed from the start; the red bulk lies between the two)
        red_planes = np.flatnonzero((only_red | ~seg_has_fluid).all(axis=(1, 2)))
        front_planes.append(int(red_planes.max()) if red_planes.size else -1)

        assert mixed_planes >= 1
    if medium.startswith("porous"):
        assert front_planes[-1] >= 0 and front_planes[0] - front_planes[-1] >= 40, front_planes       # the front came down by more than 40 planes
    else:
        assert front_planes[0] > front_planes[1], front_planes
    # ... and left single-colour segments of the OTHER colour behind: row segments that started pure red and are flagged pure blue
    # now, i.e. went red -> mixed -> blue
    red_at_start = np.where(fluid, rB == 0.0, True).reshape(nz, ny, nx // 64, 64).all(axis=3)
    turned = int((red_at_start & (want == 2) & seg_has_fluid).sum())
    if medium.startswith("open"):
        assert turned >= 50, turned
    # (in the porous medium the grains are red-wetting: a red film stays on them, no segment behind the front turns pure blue)
    for c in runs.values():
        c.close()
