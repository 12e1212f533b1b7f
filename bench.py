#!/usr/bin/env python3
"""bench.py -- headline benchmark: MLUPS of the fused LBM time step on MI355X.

    python bench.py --gpus N --steps K --warmup W [--workload c2] [--size NX NY]

A "step" is one full lattice time step (stream + BCs + colour gradient + CSF force + MRT
collision + recolouring) over the whole synthetic domain.  At N=1 the workload is
BASELINE.json configs[1]: CSF colour-gradient D2Q9 MRT, 1024 x 1024 capillary
(SimpleGeometry walls, parameters of IniFiles/RKtwophasesetup2D.ini).  2-D configs do not
shard profitably ("replicas only"), so for N>1 each rank runs an
independent replica of the same domain (weak scaling) and `value` is the aggregate.

Prints ONE JSON line on rank 0 (contract in the task statement) with `roofline` (dominant
kernel, HBM bound, HIP-event timed on the solver's own stream) and `cpu_baseline` (the C
oracle = a port of the reference algorithm, timed on the host cores of this box on a
bounded sample; baseline only, N=1 only).
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
B_ALG = {"c2": 288.0}        # algorithmic bytes per lattice update


def build_c2(nx, ny, device, variant=0):
    from openlbmpm_amd.rk2d import RK2DSolver
    from openlbmpm_amd.geometry import simple_geometry, initial_densities_rk
    dom = simple_geometry(nx, ny)
    rR, rB = initial_densities_rk(dom, False, 10, mode="intrusion")
    s = RK2DSolver(dom, dict(relax="MRT"), device=device, variant=variant)
    s.set_macro(rR, rB)
    return s, dom, rR, rB


def cpu_baseline_c2(nx, ny, target_seconds=12.0):
    """Time the oracle (C restatement of the reference algorithm, OpenMP over nodes) on a
    bounded sample of the same workload: same domain and parameters, a few steps."""
    from oracle.rk import RKOracle
    from openlbmpm_amd.geometry import simple_geometry, initial_densities_rk
    dom = simple_geometry(nx, ny)
    rR, rB = initial_densities_rk(dom, False, 10, mode="intrusion")
    o = RKOracle(dom, dict(relax="MRT"), rR, rB)
    o.run(1)                                   # touch memory / warm up
    t0 = time.perf_counter(); o.run(2); dt = (time.perf_counter() - t0) / 2
    n = max(2, min(200, int(target_seconds / max(dt, 1e-6))))
    t0 = time.perf_counter(); o.run(n); el = time.perf_counter() - t0
    mlups = o.N * n / el / 1e6
    return dict(value=round(mlups, 3), unit="MLUPS", cores=o.threads(), kind="port",
                sample="%dx%d capillary, %d steps of the C oracle (oracle/rk_oracle.c, OpenMP), %.1f s"
                       % (nx, ny, n, el))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=2000)
    ap.add_argument("--warmup", type=int, default=200)
    ap.add_argument("--workload", default="c2")
    ap.add_argument("--size", type=int, nargs=2, default=None, metavar=("NX", "NY"))
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--variant", type=int, default=0, help="0 fused (default), 1 split-3 schedule")
    args = ap.parse_args()

    import torch
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if args.gpus != world and world > 1:
        raise SystemExit("--gpus %d but WORLD_SIZE=%d" % (args.gpus, world))
    dist = None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        torch.cuda.set_device(local_rank)
        dist.init_process_group(backend="nccl", device_id=torch.device("cuda", local_rank))
    assert torch.cuda.is_available(), "bench.py needs a GPU (the HIP library is the only compute path)"
    torch.cuda.set_device(local_rank)

    nx, ny = args.size if args.size else (1024, 1024)
    if args.workload != "c2":
        raise SystemExit("unknown workload %r" % args.workload)
    solver, dom, rR0, rB0 = build_c2(nx, ny, local_rank, args.variant)
    nfluid = solver.num_fluid_nodes

    def barrier():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    solver.step(args.warmup)
    solver.sync()
    barrier()
    t0 = time.perf_counter()
    ms_total, ms_dom = solver.step_timed(args.steps)     # HIP events on the solver's stream
    solver.sync()
    barrier()
    wall = time.perf_counter() - t0
    if dist is not None:
        t = torch.tensor([wall], dtype=torch.float64, device="cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        wall = float(t.item())
    # sanity: the state must still be finite after the timed steps
    rho = solver.get("rhoR") + solver.get("rhoB")
    assert np.isfinite(rho).all(), "non-finite density after the timed run"
    m0 = float((rR0 + rB0).sum())
    assert abs(float(rho.sum()) - m0) / m0 < 1e-2, "mass drifted: the timed run did not do real work"

    if rank == 0:
        mlups = nfluid * args.steps * world / wall / 1e6
        dom_ms_per_launch = ms_dom / args.steps
        achieved = B_ALG[args.workload] * nfluid / (dom_ms_per_launch * 1e-3) / 1e9
        out = {
            "metric": "MLUPS (million lattice updates/s)", "value": round(mlups, 2), "unit": "MLUPS",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(wall * 1e3 / args.steps, 6), "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": "f64", "data": "synthetic",
            "config": {"workload": "c2: CSF colour-gradient D2Q9 MRT, %dx%d capillary "
                                   "(SimpleGeometry walls, RKtwophasesetup2D.ini parameters, "
                                   "red intruding from the top quarter)" % (nx, ny),
                       "fluid_nodes": nfluid, "lattice_nodes": nx * ny,
                       "parallelism": "replicas x%d" % world if world > 1 else "1 gpu",
                       "kernel_schedule": "fused" if args.variant == 0 else "split-3",
                       "device_ms_per_step_hip_events": round(ms_total / args.steps, 6)},
            "roofline": {"bound": "hbm", "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS,
                         "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBS, 4), "traffic": None,
                         "kernel": solver.dominant_kernel,
                         "avg_launch_ms": round(dom_ms_per_launch, 6),
                         "algorithmic_bytes_per_launch": B_ALG[args.workload] * nfluid},
        }
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline_c2(nx, ny)
        print(json.dumps(out), flush=True)
    solver.close()
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
