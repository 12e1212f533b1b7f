#!/usr/bin/env python3
"""bench.py -- headline benchmark: MLUPS of the LBM time step on MI355X.

    python bench.py --gpus N --steps K --warmup W [--workload c5|c2|c3|c4|csf3d] [--size ...]

A "step" is one full lattice time step (stream + boundary planes + colour gradient / forces +
collision + recolouring) over the whole synthetic domain.  MLUPS counts FLUID-node updates.

Workloads (BASELINE.json configs):
  c5 (default)  D3Q19 colour gradient, 512^3 synthetic porous medium -- the configuration the
                metric "MLUPS at 1/2/4/8 GPUs; % of HBM roofline" is quoted on.  It fits one GPU
                (57 GB: fluid cells only are stored) and is z-slab decomposed over N GPUs with RCCL point-to-point halo
                exchange (strong scaling: the global 512^3 is fixed as N grows).
  c2            CSF colour-gradient D2Q9 MRT, 1024^2 capillary (configs[1]); N>1 = replicas.
  c3            explicit-forcing Shan-Chen D2Q9 MRT, 2048^2 porous (configs[2]); N>1 = replicas.
  c4            CSF colour gradient + D2Q5 tracer, 2048^2 porous (configs[3]); N>1 = replicas.
At N=1 the c2, c3 and c4 figures are measured too and reported under "secondary" in the same line.

Prints ONE JSON line on rank 0 with `roofline` (dominant kernel, HBM bound, HIP-event timed on the
stream the kernel runs on) and, at N=1, `cpu_baseline` (the C oracle -- a port of the reference
algorithm -- on this box's host cores, bounded sample; baseline only).
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0        # MI355X HBM3E spec peak (guides/MI355X_MICROARCH.md)
B_ALG = {"c1": 288.0, "c2": 288.0, "c2p": 288.0, "c3": 288.0, "c4": 368.0, "c5": 608.0, "csf3d": 608.0}   # algorithmic bytes / lattice update
SEED = 20260928


def pmc_traffic(kernel, workload_tag):
    """HBM bytes per launch of `kernel` from the committed rocprofv3 PMC passes
    (profiles/pmc_traffic.json), or None when no profile matches this exact workload."""
    try:
        with open(os.path.join(ROOT, "profiles", "pmc_traffic.json")) as fh:
            recs = json.load(fh)["kernels"]
        rec = recs.get(kernel)
        if rec and rec.get("workload") != workload_tag:      # other template instance of the same kernel (tracer on/off)
            rec = next((r for k, r in recs.items() if k.startswith(kernel) and r.get("workload") == workload_tag), None)
    except (OSError, ValueError, KeyError):
        return None
    if not rec or rec.get("workload") != workload_tag:
        return None
    return rec["traffic_bytes_per_launch"]


def under_rocprof():
    """this process is itself being profiled (tools/profile_round.sh, the driver's own rocprofv3 run): no nested profiler"""
    return any(k.startswith("ROCPROF") or k == "ROCP_TOOL_LIBRARIES" for k in os.environ) or "rocprofiler" in os.environ.get("LD_PRELOAD", "")


_LIVE = {"ok": True}        # one failed or overrunning counter pass and this run asks for no further one (the committed profile serves)


def live_pmc_traffic(kernel, child_args, timeout=150, calib="calib_pull19_b64"):
    if not _LIVE["ok"]:
        return None
    out = _live_pmc_traffic(kernel, child_args, timeout, calib)
    _LIVE["ok"] = out is not None
    return out


def _live_pmc_traffic(kernel, child_args, timeout, calib="calib_pull19_b64"):
    """HBM bytes per launch of `kernel` counted IN THIS RUN: two short rocprofv3 passes (`--kernel-trace --pmc FETCH_SIZE`, then
    `--pmc WRITE_SIZE`: one counter group per pass and no other trace domain beside it, as the guide's HBM section prescribes) over a few
    steps of this same workload in a child process.  FETCH_SIZE is scaled by the factor the SAME pass measures on the calibration kernel
    of the lattice kernel's access width (calib_pull19_b64, 19 planes of 48 MiB pulled 8 bytes per lane: the child's stream test launches
    it), WRITE_SIZE is taken as reported (measured factor 1.000 in every committed profile).  Returns a dict, or None with nothing
    changed when rocprofv3 is absent, this process is already under a profiler, or a pass fails or runs over its time."""
    import glob
    import shutil
    import signal
    import sqlite3
    import subprocess
    import tempfile
    exe = shutil.which("rocprofv3") or ("/opt/rocm/bin/rocprofv3" if os.path.exists("/opt/rocm/bin/rocprofv3") else None)
    if exe is None or under_rocprof():
        return None
    tmp = tempfile.mkdtemp(prefix="lbmpm_pmc_", dir="/tmp")
    got = {}
    try:
        for counter in ("FETCH_SIZE", "WRITE_SIZE"):
            d = os.path.join(tmp, counter)
            cmd = [exe, "--kernel-trace", "--pmc", counter, "-d", d, "-o", "x", "--", sys.executable, os.path.abspath(__file__)] + child_args
            p = subprocess.Popen(cmd, cwd="/tmp", env=dict(os.environ, TMPDIR="/tmp"), stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL,
                                 start_new_session=True)
            try:
                rc = p.wait(timeout=timeout)
            except subprocess.TimeoutExpired:
                os.killpg(p.pid, signal.SIGKILL)        # the process group this call started, nothing else
                p.wait()
                return None
            dbs = glob.glob(os.path.join(d, "**", "x_results.db"), recursive=True)
            if rc != 0 or not dbs:
                return None
            rows = sqlite3.connect(dbs[0]).execute(
                "select name, count(*), avg(v) from (select name, dispatch_id, sum(counter_value) as v from pmc_events "
                "where counter_name = ? group by name, dispatch_id) group by name", (counter,)).fetchall()
            lattice = [(n, a) for name, n, a in rows if kernel in name]
            if not lattice:
                return None
            got[counter] = {"launches": sum(n for n, _ in lattice), "kb": sum(n * a for n, a in lattice) / sum(n for n, _ in lattice),
                            "calib_kb": next((a for name, _n, a in rows if calib in name), None)}
    except (OSError, sqlite3.Error):
        return None
    finally:
        shutil.rmtree(tmp, ignore_errors=True)
    true_calib = 19.0 * (48 << 20)
    ck = got["FETCH_SIZE"]["calib_kb"]
    factor = true_calib / (ck * 1024.0) if ck else 2.0
    return {"traffic": factor * got["FETCH_SIZE"]["kb"] * 1024.0 + got["WRITE_SIZE"]["kb"] * 1024.0,
            "fetch_size_kb": round(got["FETCH_SIZE"]["kb"], 1), "write_size_kb": round(got["WRITE_SIZE"]["kb"], 1),
            "fetch_factor": round(factor, 4), "fetch_factor_from": calib + " in the same pass" if ck else "the guide's gfx950 correction (no calibration row)",
            "launches_counted": got["FETCH_SIZE"]["launches"]}


def measured_hbm(device):
    """copy / read-only GB/s of this device over 2 x 4 GiB (context for roofline.frac, which is quoted
    against the 8 TB/s specification)"""
    import ctypes as C
    from openlbmpm_amd import _lib
    a, b, c = C.c_double(0), C.c_double(0), C.c_double(0)
    if _lib.lib().lbmpm_hbm_stream_test(int(device), 4 << 30, 5, C.byref(a), C.byref(b), C.byref(c)) != 0:
        return None
    return {"copy_GBs": round(a.value, 1), "read_GBs": round(b.value, 1), "inplace_update_GBs": round(c.value, 1),
            "bytes_per_buffer": 4 << 30}


# ----------------------------------------------------------------------------- workloads
def build_c2(nx, ny, device):
    from openlbmpm_amd.rk2d import RK2DSolver
    from openlbmpm_amd.geometry import simple_geometry, initial_densities_rk
    dom = simple_geometry(nx, ny)
    rR, rB = initial_densities_rk(dom, False, 10, mode="intrusion")
    s = RK2DSolver(dom, dict(relax="MRT"), device=device)
    s.set_macro(rR, rB)
    return s, float((rR + rB).sum()), lambda: float((s.get("rhoR") + s.get("rhoB")).sum())


def build_c2p(nx, ny, device):
    """the c2 lattice with SurfaceTensionType = 'Perturbation' (the 2-D twin of the c5 model): rk2dp_fused, one launch per step"""
    from openlbmpm_amd.rk2d import RK2DSolver
    from openlbmpm_amd.geometry import simple_geometry, initial_densities_rk
    dom = simple_geometry(nx, ny)
    rR, rB = initial_densities_rk(dom, False, 10, mode="intrusion")
    s = RK2DSolver(dom, dict(relax="MRT"), device=device, perturbation=dict(AkR=0.007, AkB=0.009, solidPhi=0.5))
    s.set_macro(rR, rB)
    return s, float((rR + rB).sum()), None


def build_c3(nx, ny, device):
    from openlbmpm_amd.sc2d import SC2DSolver
    from openlbmpm_amd.geometry import porous_disks, image_domain
    img = porous_disks(nx, ny - 40, porosity=0.65, rmin=6.0, rmax=20.0, seed=SEED)
    dom = image_domain(img, 20, 0.5)
    ny2 = dom.shape[0]
    ii = np.arange(ny2)[:, None] + np.zeros(dom.shape, dtype=np.int64)
    lower = ii < ny2 - 20
    fluid = dom == 1
    r0 = np.where(fluid & lower, 1.0, 0.0) + np.where(fluid & ~lower, 0.02, 0.0)
    r1 = np.where(fluid & lower, 0.02, 0.0) + np.where(fluid & ~lower, 1.0, 0.0)
    s = SC2DSolver(dom, dict(inter="EFS", relax="MRT", outlet="Dirichlet"), device=device)
    s.set_density(r0, r1)
    return s, float((r0 + r1).sum()), None


def build_c4(nx, ny, device):
    """CSF colour gradient D2Q9 MRT + one D2Q5-MRT tracer (D = 1/6, beta = 1), porous image"""
    from openlbmpm_amd.rk2d import RK2DSolver
    from openlbmpm_amd.geometry import porous_disks, image_domain, initial_densities_rk
    img = porous_disks(nx, ny - 20, porosity=0.65, rmin=6.0, rmax=20.0, seed=SEED)
    dom = image_domain(img, 10, 0.5)
    rR, rB = initial_densities_rk(dom, True, 10)
    s = RK2DSolver(dom, dict(relax="MRT"), device=device)
    s.set_macro(rR, rB)
    s.configure_tracers(diffX=(1. / 6.,), diffY=(1. / 6.,), beta=(1.0,), inlet_conc=(1.0,))
    s.set_tracer(0, np.where(rB > 0, 0.5, 0.0))
    return s, float((rR + rB).sum()), lambda: float((s.get("rhoR") + s.get("rhoB")).sum())


C1_PAR = dict(inter="ShanChen", relax="SRT", tau0=1.0, tau1=1.0, G=3.8, Gs0=-0.40, Gs1=0.40, outlet="Periodic", vy0=0.0, vy1=0.0)


def c1_fields(n=128, radius=20):
    """BASELINE configs[0]: original Shan-Chen static droplet, fully periodic box (shanchen2D.ini parameters)"""
    yy, xx = np.mgrid[0:n, 0:n]
    inside = (xx - n / 2) ** 2 + (yy - n / 2) ** 2 <= radius * radius
    return np.ones((n, n), dtype=np.uint8), np.where(inside, 1.0, 0.06), np.where(inside, 0.06, 1.0)


def build_c1(nx, ny, device):
    from openlbmpm_amd.sc2d import SC2DSolver
    dom, r0, r1 = c1_fields(nx)
    s = SC2DSolver(dom, C1_PAR, device=device)
    s.set_density(r0, r1)
    return s, float((r0 + r1).sum()), None


def build_csf3d(size, device, relax, state="initial"):
    """the c5 lattice under the 3-D CSF model (lbmpm_rk3dcsf_*), RKtwophasesetup2D.ini's surface-tension parameters"""
    from openlbmpm_amd.rk3dcsf import RK3DCSFSolver
    dom = c5_domain(size)
    dom[0] = dom[1]; dom[-1] = dom[-2]
    rR, rB = c5_state(dom, 0, size[2], state)
    s = RK3DCSFSolver(dom, dict(relax=relax, tauB=0.8), device=device)
    s.set_macro(rR, rB)
    return s, None, None


class _CSF3DRanks:
    """--workload csf3d --gpus N: the same lattice cut into one z-slab per rank (rk3dcsf.RK3DCSFDistributed: three face messages per step over
    torch.distributed), strong scaling like the c5 line.  Timed by the wall clock between barriers; no per-kernel events."""

    def __init__(self, size, device, relax, state):
        from openlbmpm_amd.rk3dcsf import RK3DCSFDistributed
        dom = c5_domain(size)
        dom[0] = dom[1]; dom[-1] = dom[-2]
        rR, rB = c5_state(dom, 0, size[2], state)
        self.d = RK3DCSFDistributed(dom, dict(relax=relax, tauB=0.8), device=device)
        self.d.set_macro(rR, rB)
        self.num_fluid_nodes = int((dom == 1).sum())        # of the whole lattice
        self.step, self.sync, self.close, self.dominant_kernel = self.d.step, self.d.sync, self.d.close, "csf3d_collide_deep"

    def step_timed(self, n):
        import time as _t
        self.d.sync()
        t0 = _t.perf_counter()
        self.d.step(n); self.d.sync()
        ms = (_t.perf_counter() - t0) * 1e3
        return ms, ms


def c5_domain(n):
    from openlbmpm_amd.geometry import porous_spheres
    nx, ny, nz = n
    return porous_spheres(nx, ny, nz, porosity=0.65, rmin=6.0, rmax=20.0, seed=SEED, nbuf=10)


def c5_densities(dom_slab, z0, nz_global, nbuf=10):
    zz = (np.arange(dom_slab.shape[0]) + z0)[:, None, None]
    fluid = dom_slab == 1
    red = zz < nz_global - nbuf
    return np.where(fluid & red, 1.0, 0.0), np.where(fluid & ~red, 1.0, 0.0)


C5_STATES = {
    "initial": "SURVEY 8d's initial state: red below the 10 blue buffer planes (most row segments hold one colour)",
    "mixed": "both colours in every fluid cell (rho_R = rho_B = 0.5): no single-colour row segment anywhere -- every record moved, "
             "every rim cell pulled; the kernel's worst case",
    "graded": "red below, blue above, the middle 30 % of the planes a linear mixture of the two",
}


def c5_state(dom_slab, z0, nz_global, state="initial", nbuf=10):
    """initial densities of the c5 legs (C5_STATES); `initial` is the state SURVEY.md 8d prescribes"""
    if state == "initial":
        return c5_densities(dom_slab, z0, nz_global, nbuf)
    fluid = dom_slab == 1
    if state == "mixed":
        return np.where(fluid, 0.5, 0.0), np.where(fluid, 0.5, 0.0)
    if state == "graded":
        zz = (np.arange(dom_slab.shape[0]) + z0)[:, None, None]
        w = np.clip((zz - 0.35 * nz_global) / (0.3 * nz_global), 0.0, 1.0) * np.ones(dom_slab.shape)
        return np.where(fluid, 1.0 - w, 0.0), np.where(fluid, w, 0.0)
    raise ValueError(state)


def c5_bytes_moved(storage, per_launch_ms, traffic):
    """what one launch moves: by the storage's own count at the end of the run (owned cells: populations + records of unflagged row
    segments; rim / halo re-reads not included) and, when a committed profile matches the workload, by the hardware counters"""
    rate = lambda b: round(b / (per_launch_ms * 1e-3) / 1e9, 1)
    return {"doubles_stored_per_cell": storage["doubles_per_cell"],
            "cells_in_single_colour_rows": storage["cells_in_flagged_rows"], "fluid_cells": storage["fluid_cells"],
            "storage_bytes_per_launch": storage["bytes_per_step"], "storage_GBs": rate(storage["bytes_per_step"]),
            "storage_frac": round(rate(storage["bytes_per_step"]) / HBM_PEAK_GBS, 4),
            "counted_bytes_per_launch": traffic, "counted_GBs": rate(traffic) if traffic else None,
            "counted_frac": round(rate(traffic) / HBM_PEAK_GBS, 4) if traffic else None}


def time_solver_2d(solver, steps, warmup):
    solver.step(warmup)
    solver.sync()
    t0 = time.perf_counter()
    ms_total, ms_dom = solver.step_timed(steps)
    solver.sync()
    return time.perf_counter() - t0, ms_total, ms_dom


# ----------------------------------------------------------------------------- CPU baselines
def cpu_team():
    """every core this process may use runs the timed CPU leg; what the box has and what was used goes in the line"""
    import oracle
    n = oracle.usable_cores()
    L = oracle.lib(threads=n)
    return dict(cores=int(L.rk_oracle_threads()), cores_box=os.cpu_count(), cores_usable=n, cpu_model=oracle.cpu_model())


def cpu_baseline_c2(nx, ny, target_seconds=10.0):
    from oracle.rk import RKOracle
    from openlbmpm_amd.geometry import simple_geometry, initial_densities_rk
    dom = simple_geometry(nx, ny)
    rR, rB = initial_densities_rk(dom, False, 10, mode="intrusion")
    team = cpu_team()
    o = RKOracle(dom, dict(relax="MRT"), rR, rB)
    o.run(1)
    t0 = time.perf_counter(); o.run(2); dt = (time.perf_counter() - t0) / 2
    n = max(2, min(200, int(target_seconds / max(dt, 1e-6))))
    t0 = time.perf_counter(); o.run(n); el = time.perf_counter() - t0
    return dict(value=round(o.N * n / el / 1e6, 3), unit="MLUPS", kind="port",
                sample="c2 %dx%d capillary, %d steps of oracle/rk_oracle.c (OpenMP, %d threads), %.1f s" % (nx, ny, n, team["cores"], el), **team)


def cpu_baseline_c5(relax, edge=128, target_seconds=12.0):
    from oracle.rk3d import RK3DOracle
    from oracle import lib
    from openlbmpm_amd.geometry import porous_spheres
    dom = porous_spheres(edge, edge, edge, porosity=0.65, rmin=6.0, rmax=20.0, seed=SEED, nbuf=10)
    rR, rB = c5_densities(dom, 0, edge)
    team = cpu_team()
    o = RK3DOracle(dom, rR, rB, dict(relax=relax))
    nfl = int(dom.sum())
    o.run(1)
    t0 = time.perf_counter(); o.run(1); dt = time.perf_counter() - t0
    n = max(1, min(2000, int(target_seconds / max(dt, 1e-6))))       # about 12 s of CPU work
    t0 = time.perf_counter(); o.run(n); el = time.perf_counter() - t0
    return dict(value=round(nfl * n / el / 1e6, 3), unit="MLUPS", kind="port",
                sample="c5 model (%s%s) on a %d^3 porous sample (same generator/parameters), %d steps of "
                       "oracle/rk3d_oracle.c (OpenMP, %d threads = every usable core), %.1f s"
                       % (relax, ": relaxation by four moment projections + odd/even split, not 19x19 products" if relax == "MRT" else "",
                          edge, n, team["cores"], el), **team)


def cpu_baseline_csf3d(relax, edge=96, target_seconds=8.0):
    """the 3-D CSF model's oracle (oracle/rk3d_csf_oracle.c, OpenMP) on a bounded sample of the leg's workload"""
    from oracle.rk3dcsf import RK3DCSFOracle
    from openlbmpm_amd.geometry import porous_spheres
    dom = porous_spheres(edge, edge, edge, porosity=0.65, rmin=6.0, rmax=20.0, seed=SEED, nbuf=10)
    dom[0] = dom[1]; dom[-1] = dom[-2]
    rR, rB = c5_densities(dom, 0, edge)
    team = cpu_team()
    o = RK3DCSFOracle(dom, rR, rB, dict(relax=relax, tauB=0.8))
    nfl = int(dom.sum())
    o.run(1)
    t0 = time.perf_counter(); o.run(1); dt = time.perf_counter() - t0
    n = max(1, min(2000, int(target_seconds / max(dt, 1e-6))))
    t0 = time.perf_counter(); o.run(n); el = time.perf_counter() - t0
    return dict(value=round(nfl * n / el / 1e6, 3), unit="MLUPS", kind="port",
                sample="3-D CSF model (%s) on a %d^3 porous sample (same generator / parameters), %d steps of oracle/rk3d_csf_oracle.c "
                       "(OpenMP, %d threads; MRT by 19 x 19 products as the reference's 2-D kernel does it), %.1f s" % (relax, edge, n, team["cores"], el), **team)


def cpu_baseline_simple_d2q9(target_seconds=6.0):
    """the 'repo's own CPU path' line: configs[0] in the shape of the reference's SimpleD2Q9 (whole-array NumPy,
    one thread), see oracle/simple_d2q9.py"""
    from oracle.simple_d2q9 import SimpleD2Q9SC
    dom, r0, r1 = c1_fields(128)
    s = SimpleD2Q9SC(dom, r0, r1, tau=(1.0, 1.0), G=3.8, Gs=(-0.40, 0.40))
    s.run(2)
    t0 = time.perf_counter(); s.run(5); dt = (time.perf_counter() - t0) / 5
    n = max(5, min(400, int(target_seconds / max(dt, 1e-6))))
    t0 = time.perf_counter(); s.run(n); el = time.perf_counter() - t0
    assert np.isfinite(s.rho[0]).all()
    import oracle
    return dict(value=round(128 * 128 * n / el / 1e6, 3), unit="MLUPS", cores=1, cores_box=os.cpu_count(), cpu_model=oracle.cpu_model(), kind="port",
                sample="configs[0] (original Shan-Chen, 128x128 periodic static droplet), %d steps of oracle/simple_d2q9.py "
                       "(NumPy, SimpleD2Q9-shaped; the reference's own CPU loop does not run), %.1f s" % (n, el))


# ----------------------------------------------------------------------------- main
def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=None)
    ap.add_argument("--warmup", type=int, default=None)
    ap.add_argument("--workload", default="c5", choices=["c5", "c2", "c3", "c4", "csf3d"],
                    help="csf3d: the c5 lattice under the 3-D CSF model (not a BASELINE config; one GPU)")
    ap.add_argument("--size", type=int, nargs="+", default=None, help="c5: NX NY NZ; c2/c3: NX NY")
    ap.add_argument("--relax", default="MRT", choices=["MRT", "SRT"],
                    help="c5 relaxation: BASELINE.json names the MRT configuration; the shipped ini says 'SRT' with ';;MRT' beside it")
    ap.add_argument("--c5-state", default="initial", choices=sorted(C5_STATES),
                    help="c5 initial condition of the primary line (default: SURVEY 8d's); the other two run as secondary legs at N = 1")
    ap.add_argument("--no-c5-legs", action="store_true", help="N = 1: skip the secondary c5 legs (other states / other relaxation)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-secondary", action="store_true")
    ap.add_argument("--no-live-traffic", action="store_true", help="N = 1, c5: skip the two short rocprofv3 counter passes that count this run's HBM bytes "
                                                                   "(roofline.traffic then comes from the committed profile)")
    ap.add_argument("--no-calibration", action="store_true", help="N > 1: keep the equal-fluid-cells cuts (no measured re-cut of the slabs)")
    args = ap.parse_args()

    import torch
    if args.gpus < 1:
        raise SystemExit("--gpus must be >= 1")
    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        # `python bench.py --gpus N` without a launcher: start the N ranks ourselves (one process per GPU, torch.distributed.run on
        # this node) and hand over; the rank-0 child prints the one JSON line
        rehearsal = os.environ.get("LBMPM_DIST_BACKEND", "nccl") != "nccl"
        have = torch.cuda.device_count() if torch.cuda.is_available() else 0
        if have < args.gpus and not rehearsal:
            raise SystemExit("bench.py --gpus %d: this node shows %d GPU(s); one rank per GPU is needed "
                             "(LBMPM_DIST_BACKEND=gloo rehearses the N-rank path on fewer GPUs, not a measurement)" % (args.gpus, have))
        import socket
        import subprocess
        with socket.socket() as sk:
            sk.bind(("127.0.0.1", 0))
            port = sk.getsockname()[1]
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus), "--master-addr", "127.0.0.1",
               "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
        raise SystemExit(subprocess.call(cmd, env=dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if args.gpus != world:
        raise SystemExit("--gpus %d but WORLD_SIZE=%d" % (args.gpus, world))
    assert torch.cuda.is_available(), "bench.py needs a GPU (the HIP library is the only compute path)"
    local_rank = local_rank % torch.cuda.device_count()    # (lets a 1-GPU box rehearse the N>1 code path)
    torch.cuda.set_device(local_rank)
    dist = None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        backend = os.environ.get("LBMPM_DIST_BACKEND", "nccl")     # "gloo": 1-GPU rehearsal of the N>1 path
        if backend == "nccl":
            dist.init_process_group(backend="nccl", device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(backend=backend)

    def barrier():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    wl = args.workload
    out = None
    if wl == "c5":
        from openlbmpm_amd.rk3d import RK3DDistributed, RK3DSlab
        from openlbmpm_amd.slab import partition_z
        size = tuple(args.size) if args.size else (512, 512, 512)
        steps = args.steps if args.steps is not None else 100       # (SURVEY.md 8d asks for 500: profiles/rNN_bench_c5_steps500.json)
        warmup = args.warmup if args.warmup is not None else max(1, steps // 10)
        dom = c5_domain(size)
        nz = size[2]
        nfluid_global = int(dom.sum())
        z0, nzl = RK3DDistributed.partition(dom, world)[rank]       # equal fluid cells per rank
        rR, rB = c5_state(dom[z0:z0 + nzl], z0, nz, args.c5_state)
        m0_local = float((rR + rB).sum())
        # the K timed steps run as (up to) five back-to-back windows: the line carries their median / min / max beside the whole-run value
        nwin = min(5, steps)
        win_steps = [steps // nwin + (1 if i < steps % nwin else 0) for i in range(nwin)]
        win_ms = []
        if world == 1:
            slab = RK3DSlab(dom, 0, nz, dict(relax=args.relax), device=local_rank)
            slab.set_density(rR, rB)
            del rR, rB
            slab.step_single(warmup)
            slab.sync(); barrier()
            t0 = time.perf_counter()
            ms_total = ms_dom = 0.0
            for n_w in win_steps:
                a, b = slab.step_timed(n_w)         # HIP events on the kernel's stream: whole window / sum over the launches
                ms_total += a; ms_dom += b; win_ms.append(a)
            slab.sync(); barrier()
            wall = time.perf_counter() - t0
            storage = slab.storage_info()
            slab.phase_field(diagnostics=True)
            rho = slab.get("rhoR") + slab.get("rhoB")
            nfl_local, dom_kernel = slab.num_fluid_nodes, slab.dominant_kernel
            slab.close()
        else:
            d = RK3DDistributed(dom, dict(relax=args.relax), device=local_rank)
            d.slab.set_density(rR, rB)
            del rR, rB
            partition = "equal fluid cells per rank"
            if not args.no_calibration:
                # two measured re-cuts (untimed set-up): equal fluid cells leave the rank that holds the colour interface ~10 % behind, and
                # the end ranks (one boundary range instead of two) ahead; the second re-cut corrects what the first one's uniform cost per
                # rank could not see
                for _recut in range(2):
                    d.step(2)
                    cost = d.calibrated_plane_cost(8)
                    d.close()
                    z0, nzl = RK3DDistributed.partition(dom, world, plane_cost=cost)[rank]
                    rR, rB = c5_state(dom[z0:z0 + nzl], z0, nz, args.c5_state)
                    m0_local = float((rR + rB).sum())
                    d = RK3DDistributed(dom, dict(relax=args.relax), device=local_rank, plane_cost=cost)
                    d.slab.set_density(rR, rB)
                    del rR, rB
                partition = "equal measured step time per rank (8 timed steps on the equal-fluid-cells cuts, re-cut, 8 timed steps, re-cut)"
            d.step(warmup)
            d.sync(); barrier()
            t0 = time.perf_counter()
            tms = []
            for n_w in win_steps:          # one call into the library per window; per-phase HIP events on the slab's streams
                t1 = time.perf_counter()
                d.step(n_w, timed=True)
                d.sync()
                win_ms.append((time.perf_counter() - t1) * 1e3)
                tms.append(d.timing())
            barrier()
            wall = time.perf_counter() - t0
            tm = dict(tms[-1])             # phase times: step-weighted averages over the windows
            tm["steps"] = sum(t["steps"] for t in tms)
            for key in ("step_ms", "interior_ms", "exchange_chain_ms", "boundary_ms", "exchange_exposed_ms"):
                tm[key] = sum(t[key] * t["steps"] for t in tms) / max(tm["steps"], 1)
            storage = d.slab.storage_info()
            ms_dom = tm["step_ms"] * steps
            ms_total = wall * 1e3
            mine = dict(rank=rank, planes=[int(z0), int(z0 + nzl)], fluid_nodes=int(d.slab.num_fluid_nodes), device="cuda:%d %s" % (local_rank, torch.cuda.get_device_name(local_rank)),
                        **{k: (round(v, 5) if isinstance(v, float) else v) for k, v in tm.items()})
            per_rank = [None] * world
            dist.all_gather_object(per_rank, mine)
            d.observe()
            rho = d.slab.get("rhoR") + d.slab.get("rhoB")
            nfl_local, dom_kernel = d.slab.num_fluid_nodes, d.slab.dominant_kernel
            d.close()
        assert np.isfinite(rho).all(), "non-finite density after the timed run"
        assert abs(float(rho.sum()) - m0_local) / max(m0_local, 1.0) < 2e-2, "mass drifted: no real work done?"
        if dist is not None:
            t = torch.tensor([wall], dtype=torch.float64, device="cuda" if dist.get_backend() == "nccl" else "cpu")
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            wall = float(t.item())
        if rank == 0:
            per_launch_ms = ms_dom / steps
            by_balg = B_ALG["c5"] * nfl_local / (per_launch_ms * 1e-3) / 1e9
            win = sorted(nfluid_global * n / (ms * 1e-3) / 1e6 for n, ms in zip(win_steps, win_ms))
            ktag = dom_kernel + ("[SRT]" if args.relax == "SRT" else "") + ("" if args.c5_state == "initial" else "[%s]" % args.c5_state)
            committed = pmc_traffic(ktag, "c5 %dx%dx%d" % size) if world == 1 else None
            live = None          # counted in this run AFTER everything has been timed (deferred below): a profiler child process between two
            deferred = []        # timed legs leaves the GPU idle for seconds, and the leg that follows pays for it
            traffic = committed
            moved = c5_bytes_moved(storage, per_launch_ms, traffic)
            # roofline.achieved / frac: BYTES MOVED per launch / launch time (the counters' figure when a committed profile matches this
            # workload, else the storage's own count); SURVEY 8d's formula (608 B per update: both colour lattices read and written
            # once) is kept beside it -- it exceeds 1 because the storage moves 23 doubles per cell, not 38
            achieved = moved["counted_GBs"] if traffic else moved["storage_GBs"]
            moved["note"] = ("the compact storage keeps 19 colour-blind populations + k_R + the recolouring vector per cell (23 doubles; the "
                             "recolouring AcceleratedRKGPU2D.py:1241-1267 makes the 38 an affine image of them) and no record at all for row "
                             "segments of a single colour; storage_* = bytes of the owned cells by the storage's own count at the end of the "
                             "run (rim / halo re-reads not included), counted_* = rocprofv3 FETCH_SIZE / WRITE_SIZE per launch, each times the factor measured in the same counter passes on a calibration "
                             "kernel of this kernel's access width (profiles/pmc_traffic.json: calibration, fetch_factor; x 2 / x 1 where a profile has none)")
            out = {
                "metric": "MLUPS (million lattice updates/s)", "value": round(nfluid_global * steps / wall / 1e6, 2),
                "unit": "MLUPS", "n_gpus": world, "steps": steps, "warmup": warmup,
                "ms_per_step": round(wall * 1e3 / steps, 5), "higher_is_better": True,
                "scaling": "strong", "vs_baseline": None, "dtype": "f64", "data": "synthetic",
                "windows": {"count": len(win), "steps_each": win_steps, "median": round(win[len(win) // 2], 2), "min": round(win[0], 2),
                            "max": round(win[-1], 2), "unit": "MLUPS",
                            "clock": "HIP events on the kernel's stream" if world == 1 else "host clock of rank 0 around each window"},
                "config": {"workload": "c5: D3Q19 colour gradient (perturbation operator, %s; RKtwophasesetup3D.ini "
                                       "parameters), %dx%dx%d synthetic porous medium (spheres r 6-20, porosity 0.65, "
                                       "10 buffer planes, side walls), seed %d; initial condition: %s" % ((args.relax,) + size + (SEED, C5_STATES[args.c5_state])),
                           "fluid_nodes": nfluid_global, "lattice_nodes": int(np.prod(size)),
                           "mlups_total_lattice": round(float(np.prod(size)) * steps / wall / 1e6, 2),
                           "parallelism": ("z-slabs x%d, RCCL p2p: one face message per cut and step (5 populations, the cell record, row flags and the "
                                           "class sums the neighbour completes its halo phase field from), sent after the boundary planes, hidden "
                                           "behind the interior planes" if dom_kernel == "rk3dq_fused" else
                                           "z-slabs x%d, RCCL p2p halo (5 populations x 2 colours + phi per face)") % world if world > 1 else "1 gpu",
                           "kernel_schedule": "one fused z-marching kernel per step (pull, phase field in an LDS ring, collide, store), "
                                              + {"rk3dq_fused": "compact storage: fluid cells only, 23 doubles per cell (19 colour-blind populations "
                                                                "+ k_R + recolouring vector in an LDS tile), row segments of one colour keep no record",
                                                 "rk3dc_fused": "compact storage: fluid cells only, both colour lattices"}.get(dom_kernel, "dense storage")
                                              if "fused" in dom_kernel else "phase_field + collide (split-2)",
                           "parity": "pinned by reduction: y-uniform lattice through this kernel == captures of the reference's real D2Q9 perturbation "
                                     "driver (RKD2Q9.py:978-1223, run with the four call-site repairs listed in tests/golden/gen/make_golden_rk_pert.py: "
                                     "R3 moves calTotalFluidPDF behind collision 1, i.e. the pin is to that repaired loop) to 1e-10, SRT "
                                     "(tests/test_rk3d_reduction.py); full 3-D and MRT vs oracle/rk3d_oracle.c 1e-10; the 23-value storage vs the "
                                     "38-value kernels 1e-11 (tests/test_rk3d_gpu.py)"},
                "roofline": {"bound": "hbm", "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                             "frac": round(achieved / HBM_PEAK_GBS, 4),
                             "achieved_is": "bytes moved per launch / avg_launch_ms: " + (
                                 "hardware counters (traffic)" if traffic else "the storage's own count (no committed counter profile matches this workload)"),
                             "achieved_by_survey_balg": round(by_balg, 1), "frac_by_survey_balg": round(by_balg / HBM_PEAK_GBS, 4),
                             "frac_by_survey_balg_note": "SURVEY 8d: MLUPS x 608 B / 8 TB/s; not a bandwidth fraction for this storage (may exceed 1)",
                             "traffic": traffic,
                             "traffic_source": ("measured in THIS run, on this GPU: two rocprofv3 passes (--kernel-trace --pmc FETCH_SIZE, then WRITE_SIZE) over "
                                                "8 steps of this workload in a child process of bench.py; FETCH_SIZE x the factor the same pass measures on "
                                                "the calibration kernel of this kernel's access width") if live else
                                               ("profiles/pmc_traffic.json: rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of tools/profile_round.sh over this "
                                                "command on an MI355X box (committed; NOT measured in this run" + (
                                                    ": this process is itself under a profiler" if under_rocprof() else
                                                    ": --no-live-traffic" if args.no_live_traffic else ": the live counter passes did not complete or do not apply") + ")") if traffic else None,
                             "traffic_live": live, "traffic_committed_profile": committed,
                             "bytes_moved": moved,
                             "kernel": dom_kernel,
                             "measured_stream_ceiling": measured_hbm(local_rank) if world == 1 else None,
                             "note": None if world == 1 else "N>1: the kernel runs as boundary + interior launches on two "
                                     "streams beside the halo exchange; avg_launch_ms brackets the whole step, exchange included",
                             "avg_launch_ms": round(per_launch_ms, 5),
                             "algorithmic_bytes_per_launch": B_ALG["c5"] * nfl_local},
            }
            LIVE_TEXT = ("measured in THIS run, on this GPU: two rocprofv3 passes (--kernel-trace --pmc FETCH_SIZE, then WRITE_SIZE) over "
                         "8 steps of this workload in a child process of bench.py, after everything has been timed; FETCH_SIZE x the factor the same pass "
                         "measures on the calibration kernel of this kernel's access width")
            if world == 1 and not args.no_live_traffic and dom_kernel == "rk3dq_fused":
                def main_live(out=out, storage=storage, per_launch_ms=per_launch_ms, note=moved["note"]):
                    # the instance that runs every step but the first (template argument FIRST = false)
                    lv = live_pmc_traffic("rk3dq_fused<false", ["--steps", "6", "--warmup", "2", "--no-cpu-baseline", "--no-secondary", "--no-live-traffic",
                                                                 "--relax", args.relax, "--c5-state", args.c5_state, "--size"] + [str(v) for v in size])
                    if not lv:
                        return
                    mv = c5_bytes_moved(storage, per_launch_ms, lv["traffic"])
                    mv["note"] = note
                    r = out["roofline"]
                    r.update({"achieved": round(mv["counted_GBs"], 1), "frac": round(mv["counted_GBs"] / HBM_PEAK_GBS, 4),
                              "achieved_is": "bytes moved per launch / avg_launch_ms: hardware counters (traffic)", "traffic": lv["traffic"],
                              "traffic_source": LIVE_TEXT, "traffic_live": lv, "bytes_moved": mv})
                deferred.append(main_live)
            if world > 1:
                # what the transport really was, and where a step's time went on every rank (HIP events inside
                # lbmpm_rk3d_step_slab): exchange_exposed_ms = step - max(interior, boundary)
                out["multi_gpu"] = {"backend": dist.get_backend(), "world_size": dist.get_world_size(),
                                    "transport": per_rank[0].get("transport", "?") + {
                                        True: " -- the library moves the face messages itself (include/lbmpm.h: LBMPM_TRANSPORT_*); torch.distributed "
                                              "(%s) serves set-up and the timing reduction only" % dist.get_backend(),
                                        False: " -- torch.distributed batch_isend_irecv per face and step (%s)" % (
                                            "ncclSend / ncclRecv pairs, RCCL" if dist.get_backend() == "nccl" else
                                            "REHEARSAL on the %s backend, host-staged copies: not an RCCL measurement" % dist.get_backend())}[
                                        "callback" not in per_rank[0].get("transport", "callback")],
                                    # which transport every rank ended on, and what set-up tried before it with the verdict of each candidate
                                    # (RK3DDistributed.transport_log: connect errors, the probe's mismatch count, the watchdog's message) --
                                    # enough to diagnose a run that fell back without a second lease on the node
                                    "transport_per_rank": [r.get("transport", "?") for r in per_rank],
                                    "transport_candidates": [{"rank": r.get("rank"), "tried": r.get("transport_log", [])} for r in per_rank
                                                             if r.get("rank") == 0 or r.get("transport_log") != per_rank[0].get("transport_log")],
                                    "host_enqueue_us_per_step": [r.get("host_enqueue_us_per_step") for r in per_rank],
                                    "boundary_depth_planes": int(os.environ.get("LBMPM_RK3D_BOUNDARY", "2")),
                                    "partition": partition,
                                    "per_rank": per_rank}
            if world == 1 and not args.no_secondary:
                sec = []
                for name, build, size2 in (("c1", build_c1, (128, 128)), ("c2", build_c2, (1024, 1024)), ("c2p", build_c2p, (1024, 1024)),
                                           ("c3", build_c3, (2048, 2048)), ("c4", build_c4, (2048, 2048))):
                    s, _, _ = build(size2[0], size2[1], local_rank)
                    k = {"c1": 20000, "c2": 5000, "c2p": 5000}.get(name, 2000)    # C2 5000, C3 / C4 2000 steps (+10 % warm-up); c1: 16 k nodes,
                                                                       # ~7 us a step, the warm-up builds its hipGraph
                    if os.environ.get("LBMPM_BENCH_SECONDARY_STEPS"):    # counter passes of tools/profile_round.sh: every dispatch is slow there
                        k = int(os.environ["LBMPM_BENCH_SECONDARY_STEPS"])
                    w, mt, md = time_solver_2d(s, k, k // 10)
                    nf = s.num_fluid_nodes
                    sec.append({"workload": name, "value": round(nf * k / w / 1e6, 2), "unit": "MLUPS",
                                "ms_per_step": round(w * 1e3 / k, 5), "fluid_nodes": nf, "kernel": s.dominant_kernel,
                                "roofline_frac": round(B_ALG[name] * nf / (md / k * 1e-3) / 1e9 / HBM_PEAK_GBS, 4)})
                    if name == "c2p":
                        sec[-1]["note"] = ("not a BASELINE config: the c2 lattice with SurfaceTensionType 'Perturbation' (RKD2Q9.py:978-1223), the 2-D twin of "
                                           "the c5 model, fused into one launch per step (tests/test_rk2d_pert_gpu.py: the reference loop's captures at 1e-9)")
                    if name == "c1":
                        sec[-1]["note"] = ("configs[0] on the GPU: 64 workgroups, bound by the kernel's own latency chain (hipGraph replay of 64 steps "
                                           "removes the launch gaps).  The static-droplet Laplace test: with shanchen2D.ini's interactionFluid = 3.8 the "
                                           "pinned original Shan-Chen kernel holds no static droplet in the periodic 128 x 128 box (radius 26 is gone within "
                                           "500 steps, radius 20 leaves the centre and shows no pressure jump after 10^4: tests/test_oracle_sc.py); at G = 2.6, "
                                           "the nearest coupling at which it stays put, three radii over 10^4 steps give dp * R within +-5 % of each other on the "
                                           "CPU oracle and on this kernel (tests/test_c1_droplet_gpu.py); this line times the ini's own parameters")
                    s.close()
                # the same 3-D workload in its other states and with the other relaxation
                def c5_leg(label, relax, state, nsteps):
                    r2, b2 = c5_state(dom, 0, nz, state)
                    s3 = RK3DSlab(dom, 0, nz, dict(relax=relax), device=local_rank)
                    s3.set_density(r2, b2)
                    del r2, b2
                    s3.step_single(5); s3.sync()
                    t1 = time.perf_counter(); mt3, md3 = s3.step_timed(nsteps); s3.sync(); w3 = time.perf_counter() - t1
                    st3 = s3.storage_info()
                    nf3, dk3 = s3.num_fluid_nodes, s3.dominant_kernel
                    ktag3 = dk3 + ("[SRT]" if relax == "SRT" else "") + ("" if state == "initial" else "[%s]" % state)
                    s3.close()
                    mv = c5_bytes_moved(st3, md3 / nsteps, pmc_traffic(ktag3, "c5 %dx%dx%d" % size))
                    mv["counted_in"] = "profiles/pmc_traffic.json (committed profile)"
                    leg = {"workload": label, "value": round(nf3 * nsteps / w3 / 1e6, 2), "unit": "MLUPS",
                           "ms_per_step": round(w3 * 1e3 / nsteps, 5), "steps": nsteps, "fluid_nodes": nf3, "kernel": dk3,
                           "state": C5_STATES[state], "cells_in_single_colour_rows": mv["cells_in_single_colour_rows"],
                           "roofline_frac": mv["counted_frac"] if mv["counted_frac"] is not None else mv["storage_frac"],
                           "roofline_frac_by_survey_balg": round(B_ALG["c5"] * nf3 / (md3 / nsteps * 1e-3) / 1e9 / HBM_PEAK_GBS, 4),
                           "bytes_moved": mv}
                    if not args.no_live_traffic and dk3 == "rk3dq_fused":
                        def leg_live(leg=leg, st3=st3, ms=md3 / nsteps):
                            lv = live_pmc_traffic("rk3dq_fused<false", ["--steps", "6", "--warmup", "2", "--no-cpu-baseline", "--no-secondary", "--no-live-traffic",
                                                                         "--relax", relax, "--c5-state", state, "--size"] + [str(v) for v in size])
                            if not lv:
                                return
                            m2 = c5_bytes_moved(st3, ms, lv["traffic"])
                            m2["counted_in"] = "this run (rocprofv3 counter passes in a child process, after the timed legs)"
                            leg["bytes_moved"] = m2
                            leg["roofline_frac"] = m2["counted_frac"]
                        deferred.append(leg_live)
                    return leg
                if not args.no_c5_legs:
                    for state in sorted(C5_STATES):
                        if state != args.c5_state:
                            sec.append(c5_leg("c5 %s (%s)" % ({"mixed": "two colours in every cell", "graded": "graded (30 % of the planes mixed)",
                                                                  "initial": "initial state"}[state], args.relax), args.relax, state, 40))
                    other = "SRT" if args.relax == "MRT" else "MRT"
                    sec.append(c5_leg("c5 with %s relaxation" % other, other, args.c5_state, 30))
                    # the same lattice under the OTHER surface-tension model of the 2-D ini: the CSF loop carried to D3Q19 (a17's "CSF kappa = -div n in 3-D")
                    from openlbmpm_amd.rk3dcsf import RK3DCSFSolver
                    d2 = dom.copy(); d2[0] = d2[1]; d2[-1] = d2[-2]
                    r2, b2 = c5_state(dom, 0, nz, "initial")
                    sc3 = RK3DCSFSolver(d2, dict(relax=args.relax, tauB=0.8), device=local_rank)
                    sc3.set_macro(r2, b2)
                    del r2, b2, d2
                    sc3.step(10); sc3.sync()
                    kc = 100                    # (like the primary line: 10 + 100 steps from the drainage's initial state)
                    t1 = time.perf_counter(); mtc, mdc = sc3.step_timed(kc); sc3.sync(); wc = time.perf_counter() - t1
                    nfc = sc3.num_fluid_nodes
                    sec.append({"workload": "c5 lattice, [SurfaceTension] SurfaceTensionType = 'CSF' (%s): the 2-D CSF loop (RKD2Q9.py:1295-1490) carried to D3Q19, "
                                            "RKtwophasesetup2D.ini's parameters (sigma 0.1, contact angle 60, wetting rule 2, TauType 2)" % args.relax,
                                "value": round(nfc * kc / wc / 1e6, 2), "unit": "MLUPS", "ms_per_step": round(wc * 1e3 / kc, 5), "steps": kc, "fluid_nodes": nfc,
                                "wetting_solids": sc3.num_wetting_solids, "kernel": "csf3d_collide_deep", "collide_ms": round(mdc / kc, 5),
                                "roofline_frac_by_survey_balg": round(B_ALG["c5"] * nfc / (mtc / kc * 1e-3) / 1e9 / HBM_PEAK_GBS, 4),
                                "bulk_cells": sc3.bulk_cells,
                                "note": "not a BASELINE config (the 3-D ini carries the perturbation parameters).  Where two colours meet: phase field (pull 38), "
                                        "solid phi, gradient + wetting rule, collide (pull 38, store 38) -- the curvature needs the normal one cell around and the "
                                        "normal the phase field one cell around that.  Blocks of 256 fluid cells deep inside one colour (bulk_cells of fluid_nodes "
                                        "in the last step) skip phase field and gradient and collide the present colour alone through a table of source cells "
                                        "(19 loads, 19 stores): exact, bit-equal to the full path.  roofline_frac_by_survey_balg is over the whole step (608 B per "
                                        "update, which the bulk path does not move: it may exceed the bandwidth fraction; counted: DESIGN.md section 4).  Parity: "
                                        "oracle/rk3d_csf_oracle.c at 1e-10, pinned by reduction to the capture of the real 2-D driver (tests/test_rk3d_csf_gpu.py)"})
                    csf_leg = sec[-1]
                    if not args.no_live_traffic:
                        def csf_live(leg=csf_leg, ms=mdc / kc):
                            # bytes the bulk's collision moves, counted in a child of this run (csf3d_collide_deep runs beside the full path's
                            # launches: its own time is not separable by events; the fraction below is over the step's time, all launches)
                            lv = live_pmc_traffic("csf3d_collide_deep", ["--workload", "csf3d", "--steps", "6", "--warmup", "12", "--no-cpu-baseline", "--relax", args.relax,
                                                                         "--size"] + [str(v) for v in size], calib="calib_shift19_b64")
                            if lv:
                                leg["bulk_collision_counted"] = dict(lv, GBs_over_the_step=round(lv["traffic"] / (ms * 1e-3) / 1e9, 1),
                                                                     frac_over_the_step=round(lv["traffic"] / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4),
                                                                     note="FETCH_SIZE x the factor the same pass measures on calib_shift19_b64 (19 planes pulled "
                                                                          "through windows that do not start on a line, like the pulls of the fluid-cells-only "
                                                                          "numbering); by the kernel's own count a bulk cell moves 376 B (19 loads, 18 table words, "
                                                                          "19 stores)",
                                                                     own_count_bytes=376 * leg["bulk_cells"],
                                                                     own_count_frac_over_the_step=round(376 * leg["bulk_cells"] / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4))
                        deferred.append(csf_live)
                    # ... and with both colours in every cell: every block on the full path
                    r2, b2 = c5_state(dom, 0, nz, "mixed")
                    sc3.set_macro(r2, b2)
                    del r2, b2
                    sc3.step(3); sc3.sync()
                    t1 = time.perf_counter(); mtm, mdm = sc3.step_timed(10); sc3.sync(); wm = time.perf_counter() - t1
                    sec.append({"workload": "c5 lattice, SurfaceTensionType = 'CSF' (%s), two colours in every cell" % args.relax, "value": round(nfc * 10 / wm / 1e6, 2),
                                "unit": "MLUPS", "ms_per_step": round(wm * 1e2, 5), "steps": 10, "fluid_nodes": nfc, "kernel": "csf3d_collide", "bulk_cells": sc3.bulk_cells,
                                "roofline_frac_by_survey_balg": round(B_ALG["c5"] * nfc / (mtm / 10 * 1e-3) / 1e9 / HBM_PEAK_GBS, 4),
                                "note": "the full path everywhere: 136 doubles per cell and step by the schedule's own count (1.37 kB counted) against B_alg = 608 B"})
                    sc3.close()
                out["secondary"] = sec
            if world == 1 and not args.no_cpu_baseline:
                out["cpu_baseline"] = cpu_baseline_c5(args.relax)
                out["cpu_baseline_reference_shape"] = cpu_baseline_simple_d2q9()
                if not args.no_secondary and not args.no_c5_legs:
                    out["cpu_baseline_csf3d"] = cpu_baseline_csf3d(args.relax)
            for fn in deferred:          # the counter passes: nothing is timed after this point
                fn()
    else:
        size = tuple(args.size) if args.size else ((512, 512, 512) if wl == "csf3d" else ((1024, 1024) if wl == "c2" else (2048, 2048)))
        steps = args.steps if args.steps is not None else (100 if wl == "csf3d" else (2000 if wl == "c2" else 500))
        warmup = args.warmup if args.warmup is not None else steps // 10
        csf_ranks = wl == "csf3d" and world > 1
        if csf_ranks:
            solver, m0, mass = _CSF3DRanks(size, local_rank, args.relax, args.c5_state), None, None
        elif wl == "csf3d":
            solver, m0, mass = build_csf3d(size, local_rank, args.relax, args.c5_state)
        else:
            solver, m0, mass = {"c2": build_c2, "c3": build_c3, "c4": build_c4}[wl](size[0], size[1], local_rank)
        nfluid = solver.num_fluid_nodes
        solver.step(warmup)
        solver.sync(); barrier()
        if csf_ranks:
            solver.d.timing()                    # (the warm-up steps leave the per-stage clocks)
        t0 = time.perf_counter()
        ms_total, ms_dom = solver.step_timed(steps)
        solver.sync(); barrier()
        wall = time.perf_counter() - t0
        if mass is not None:
            m1 = mass()
            assert np.isfinite(m1) and abs(m1 - m0) / m0 < 1e-2, "mass drifted: the timed run did not do real work"
        if dist is not None:
            t = torch.tensor([wall], dtype=torch.float64, device="cuda" if dist.get_backend() == "nccl" else "cpu")
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            wall = float(t.item())
        per_rank_timing = None
        if csf_ranks:
            per_rank_timing = [None] * world
            dist.all_gather_object(per_rank_timing, dict(solver.d.timing(), rank=rank))
        if rank == 0:
            per_launch_ms = ms_dom / steps
            achieved = B_ALG[wl] * nfluid / (per_launch_ms * 1e-3) / 1e9
            desc = {"c2": "c2: CSF colour-gradient D2Q9 MRT, %dx%d capillary (SimpleGeometry walls, "
                          "RKtwophasesetup2D.ini parameters, red intruding from the top quarter)",
                    "c3": "c3: explicit-forcing Shan-Chen D2Q9 MRT (efs2D.ini parameters), %dx%d synthetic "
                          "porous image (discs r 6-20, porosity 0.65)",
                    "c4": "c4: CSF colour-gradient D2Q9 MRT + one D2Q5-MRT tracer, %dx%d synthetic porous "
                          "image (discs r 6-20, porosity 0.65)",
                    "csf3d": "not a BASELINE config: the c5 lattice %dx%dx%d under the 3-D CSF model (" + args.relax + ", state " + args.c5_state + ")"}[wl] % size
            out = {
                "metric": "MLUPS (million lattice updates/s)", "value": round(nfluid * steps * (1 if csf_ranks else world) / wall / 1e6, 2),
                "unit": "MLUPS", "n_gpus": world, "steps": steps, "warmup": warmup,
                "ms_per_step": round(wall * 1e3 / steps, 6), "higher_is_better": True, "scaling": "strong" if csf_ranks else "weak",
                "vs_baseline": None, "dtype": "f64", "data": "synthetic",
                "config": {"workload": desc, "fluid_nodes": nfluid, "lattice_nodes": int(np.prod(size)),
                           "parallelism": ("z-slabs x%d (a ring; phi, n, crossing populations as face messages)" % world if csf_ranks else "replicas x%d" % world) if world > 1 else "1 gpu",
                           "kernel_schedule": "fused", "device_ms_per_step_hip_events": round(ms_total / steps, 6)},
                "roofline": {"bound": "hbm", "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                             "frac": round(achieved / HBM_PEAK_GBS, 4),
                             "traffic": pmc_traffic(solver.dominant_kernel, "%s %s" % (wl, "x".join(str(v) for v in size))),
                             "traffic_source": "profiles/pmc_traffic.json (rocprofv3 counter passes of tools/profile_round.sh, committed; not measured in this run)",
                             "measured_stream_ceiling": measured_hbm(local_rank),
                             "kernel": solver.dominant_kernel, "avg_launch_ms": round(per_launch_ms, 6),
                             "algorithmic_bytes_per_launch": B_ALG[wl] * nfluid},
            }
            if world == 1 and not args.no_cpu_baseline and wl == "c2":
                out["cpu_baseline"] = cpu_baseline_c2(*size)
            if wl == "csf3d":
                out["config"]["kernel_schedule"] = ("per step: bookkeeping of the bulk skip, phase field / solid phi / gradient / collision of the blocks on the full "
                                                    "path, and beside them csf3d_collide_deep for the blocks deep inside one colour (19 loads through a table of source cells, 19 stores)")
                out["config"]["bulk_cells"] = None if csf_ranks else solver.bulk_cells
                if csf_ranks:
                    out["multi_gpu"] = {"backend": dist.get_backend(), "world_size": world, "per_rank": per_rank_timing,
                                        "note": "host clocks per step and stage: wait_stage = the stage's launches on the first stream + the pack have run; message = "
                                                "both faces' messages sent, received and unpacked (phi and n travel while the bulk's collision runs on the second stream; "
                                                "the populations' message is exposed)"}
                out["roofline"].update(kernel="csf3d_collide_deep", achieved_is="B_alg (608 B) x fluid nodes / step time: NOT a bandwidth fraction -- the bulk path does not move "
                                       "608 B per cell; counted bytes below when this run could count them", frac_by_survey_balg=out["roofline"]["frac"])
                lv = None if (args.no_live_traffic or world != 1) else live_pmc_traffic(
                    "csf3d_collide_deep", ["--workload", "csf3d", "--steps", "6", "--warmup", "12", "--no-cpu-baseline", "--no-live-traffic", "--relax", args.relax,
                                           "--c5-state", args.c5_state, "--size"] + [str(v) for v in size], calib="calib_shift19_b64")
                if lv:
                    gbs = lv["traffic"] / (per_launch_ms * 1e-3) / 1e9
                    out["roofline"].update(traffic=lv["traffic"], traffic_live=lv, achieved=round(gbs, 1), frac=round(gbs / HBM_PEAK_GBS, 4),
                                           achieved_is="bytes csf3d_collide_deep moved per launch (hardware counters; FETCH_SIZE x the factor of calib_shift19_b64, "
                                                       "the calibration kernel with unaligned windows, in the same pass) / the step's time (HIP events)",
                                           own_count_bytes=376 * solver.bulk_cells, own_count_frac=round(376 * solver.bulk_cells / (per_launch_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4),
                                           traffic_source="measured in THIS run: two rocprofv3 passes (FETCH_SIZE, WRITE_SIZE) over a child process; bytes of "
                                                          "csf3d_collide_deep per launch over the step's time by HIP events (that kernel runs beside the full path's "
                                                          "launches; alone it takes ~ 0.85 of the step)")
                if world == 1 and not args.no_cpu_baseline:
                    out["cpu_baseline"] = cpu_baseline_csf3d(args.relax)
        solver.close()
    if rank == 0:
        print(json.dumps(out), flush=True)
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
