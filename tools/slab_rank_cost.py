#!/usr/bin/env python3
"""GPU time of ONE rank's slab step, rank by rank, against the single-domain step (one GPU, no second process).

`tools/slabbench.py` drives k virtual ranks from one Python thread: its k = 8 figure contains ~150 host calls per lattice
step that a real run spreads over eight processes.  Here every rank of the `bench.py --gpus K` partition is built alone and
stepped through `lbmpm_rk3d_step_slab(timed)` -- the call a real rank makes -- with an exchange callback that copies the
neighbours' last face messages (device copies: real messages of the right shape, same kernels, same launch sequence, no xGMI).
HIP events give step / interior / boundary-path time per rank; sum over ranks vs the single-domain step is the GPU cost
of the decomposition, K x max over ranks vs the same is the load imbalance on top of it.

    python tools/slab_rank_cost.py [n=512] [K=8] [steps=40]        (LBMPM_K3_RELAX=SRT|MRT; SLAB_CALIBRATE=1: then re-cut the slabs by
                                                                    the measured cost per plane, as bench.py --gpus N does, and measure again)
"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import bench
from openlbmpm_amd.rk3d import RK3DSlab, RK3DDistributed

n = int(sys.argv[1]) if len(sys.argv) > 1 else 512
K = int(sys.argv[2]) if len(sys.argv) > 2 else 8
steps = int(sys.argv[3]) if len(sys.argv) > 3 else 40
par = dict(relax=os.environ.get("LBMPM_K3_RELAX", "MRT"))
dom = bench.c5_domain((n, n, n))
rR, rB = bench.c5_densities(dom, 0, n)

s = RK3DSlab(dom, 0, n, par)
s.set_density(rR, rB)
s.step_single(3)
single = min(s.step_timed(10)[0] / 10 for _ in range(3))
nf = s.num_fluid_nodes
s.close()
print("single domain: %.3f ms per step, %d fluid nodes" % (single, nf), flush=True)

only = [int(v) for v in os.environ["SLAB_RANKS"].split(",")] if os.environ.get("SLAB_RANKS") else None      # a subset of the ranks
st = torch.cuda.Stream(0)


def measure(parts):
    """every rank of `parts` resident, advancing together; returns one timing dict per rank"""
    slabs = []
    for r, (z0, nz) in enumerate(parts):          # all K ranks resident (the whole lattice is one GPU's worth of memory anyway)
        s = RK3DSlab(dom, z0, nz, par)
        s.set_density(rR[z0:z0 + nz], rB[z0:z0 + nz])
        s.use_torch_stream(st)
        slabs.append(s)
    with torch.cuda.stream(st):                    # every rank's face message once: the first round's callbacks find real messages
        for s in slabs:
            s.pack()
    # The ranks advance TOGETHER, one lbmpm_rk3d_step_slab(1) each per round: a callback then copies its neighbours' latest face messages,
    # at most one step old.  (Timing one rank alone for many steps against messages that never change fills its slab with a mixture of the
    # two colours from the faces inwards -- the phase field of the halo planes no longer matches -- and measures the kernel's worst case:
    # the first version of this tool reported 1.5 ms per rank that way.)
    acc = [dict(step_ms=0.0, interior_ms=0.0, boundary_ms=0.0, exchange_chain_ms=0.0) for _ in parts]

    def make_exchange(r):
        s, below, above = slabs[r], r > 0, r + 1 < K

        def exchange(what):
            kind = "phi" if what else "f"
            if s.buffer(kind + "_send_up") is None:
                return
            if below:
                s.buffer(kind + "_recv_below").copy_(slabs[r - 1].buffer(kind + "_send_up"))
            if above:
                s.buffer(kind + "_recv_above").copy_(slabs[r + 1].buffer(kind + "_send_down"))
        return exchange

    cbs = [make_exchange(r) for r in range(K)]
    warm = 4
    with torch.cuda.stream(st):
        for k in range(warm + steps):
            for r in range(K):
                slabs[r].step_slab(1, r > 0, r + 1 < K, cbs[r], timed=k >= warm)
                if k >= warm:
                    t = slabs[r].slab_timing()
                    for key in acc[r]:
                        acc[r][key] += t[key] / steps
    rows = []
    for r, (z0, nz) in enumerate(parts):
        t = dict(acc[r], rank=r, planes=nz, fluid=slabs[r].num_fluid_nodes)
        rows.append(t)
        print("rank %d: planes %3d  fluid %9d  step %.3f ms  interior %.3f  boundary %.3f  pack..unpack chain %.3f   (%s)" %
              (r, nz, t["fluid"], t["step_ms"], t["interior_ms"], t["boundary_ms"], t["exchange_chain_ms"], slabs[r].dominant_kernel), flush=True)
    for s in slabs:
        s.close()
    tot = sum(t["step_ms"] for t in rows)
    mx = max(t["step_ms"] for t in rows)
    print("sum over ranks %.3f ms = single x %.3f;  %d x slowest rank %.3f ms = single x %.3f" % (tot, tot / single, K, K * mx, K * mx / single))
    return rows


parts = RK3DDistributed.partition(dom, K)
rows = measure(parts)
# what RK3DDistributed.calibrated_plane_cost + partition(plane_cost=...) do in a real run: a rank's step time per owned plane, re-cut;
# SLAB_CALIBRATE = number of re-cuts (bench.py --gpus N does two)
import numpy as np
cost = None
for it in range(int(os.environ.get("SLAB_CALIBRATE", "0"))):
    new = np.zeros(n)
    for (z0, nz), t in zip(parts, rows):
        new[z0:z0 + nz] = t["step_ms"] / nz
    cost = new
    parts2 = RK3DDistributed.partition(dom, K, plane_cost=cost)
    print("re-cut %d by measured cost per plane: planes per rank %s -> %s" % (it + 1, [nz for _, nz in parts], [nz for _, nz in parts2]), flush=True)
    parts = parts2
    rows = measure(parts)
