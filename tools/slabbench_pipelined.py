#!/usr/bin/env python3
"""The slab step as the ranks of a real run execute it -- lbmpm_rk3d_step_slab, one call per rank for all steps -- with k ranks on ONE
GPU and an EXACT exchange: one host thread per rank, a barrier inside the exchange callback (every rank has enqueued its pack before
any message is copied), device copies instead of xGMI.  All ranks share the GPU, so only the total is meaningful: time per step of the
whole lattice, to be compared with the single domain (tools/slabbench.py is the same for the phase-by-phase order of RK3DCluster).

    python tools/slabbench_pipelined.py [n=512] [k=8] [steps=20]        (LBMPM_K3_RELAX=SRT|MRT)
"""
import os, sys, threading, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
import bench
from openlbmpm_amd.rk3d import RK3DSlab, RK3DDistributed

n = int(sys.argv[1]) if len(sys.argv) > 1 else 512
K = int(sys.argv[2]) if len(sys.argv) > 2 else 8
steps = int(sys.argv[3]) if len(sys.argv) > 3 else 20
par = dict(relax=os.environ.get("LBMPM_K3_RELAX", "MRT"))
dom = bench.c5_domain((n, n, n))
rR, rB = bench.c5_densities(dom, 0, n)
nf = int(dom.sum())
parts = RK3DDistributed.partition(dom, K)
st = torch.cuda.Stream(0)
slabs = []
for z0, nz in parts:
    s = RK3DSlab(dom, z0, nz, par); s.set_density(rR[z0:z0 + nz], rB[z0:z0 + nz]); s.use_torch_stream(st); slabs.append(s)
gate = threading.Barrier(K)


def rank(r, nsteps):
    s, below, above = slabs[r], r > 0, r + 1 < K

    def exchange(what):
        gate.wait()                     # every rank's pack is in the stream
        with torch.cuda.stream(st):
            if below:
                s.buffer("f_recv_below").copy_(slabs[r - 1].buffer("f_send_up"))
            if above:
                s.buffer("f_recv_above").copy_(slabs[r + 1].buffer("f_send_down"))
        gate.wait()                     # nobody packs the next message before every copy of this one is in the stream
    with torch.cuda.stream(st):
        s.step_slab(nsteps, below, above, exchange)


def run(nsteps):
    th = [threading.Thread(target=rank, args=(r, nsteps)) for r in range(K)]
    for t in th: t.start()
    for t in th: t.join()
    torch.cuda.synchronize()


run(3)
t0 = time.perf_counter()
run(steps)
dt = (time.perf_counter() - t0) / steps
print("k=%d pipelined step_slab, exact exchange: %.3f ms per step of the whole lattice (%.0f MLUPS on one GPU; per rank %.3f ms)"
      % (K, dt * 1e3, nf / dt / 1e6, dt * 1e3 / K), flush=True)
# the result is a real run's: compare a field with the single domain
with torch.cuda.stream(st):
    for r, s in enumerate(slabs):
        s.pack()
    for r, s in enumerate(slabs):
        if r > 0: s.buffer("f_recv_below").copy_(slabs[r - 1].buffer("f_send_up"))
        if r + 1 < K: s.buffer("f_recv_above").copy_(slabs[r + 1].buffer("f_send_down"))
    for r, s in enumerate(slabs):
        s.unpack(r > 0, r + 1 < K); s.phase_field(diagnostics=True)
phi = np.concatenate([s.get("phi") for s in slabs], axis=0)
for s in slabs:
    s.close()
ref = RK3DSlab(dom, 0, n, par); ref.set_density(rR, rB); ref.step_single(steps + 3); ref.phase_field(diagnostics=True)
rphi = ref.get("phi")
same = bool(np.array_equal(phi, rphi))
print("phase field equals the single domain's bit for bit:", same)
if not same:
    for name, a in (("slabs", phi), ("single", rphi)):
        bad = ~np.isfinite(a)
        if bad.any():
            per = bad.reshape(bad.shape[0], -1).sum(axis=1)
            zb = np.flatnonzero(per)
            print("  %s: %d non-finite values in planes %s" % (name, int(bad.sum()), [(int(z), int(per[z])) for z in zb[:400]]))
    d = np.abs(np.nan_to_num(phi) - np.nan_to_num(rphi))
    zz = np.flatnonzero(d.reshape(d.shape[0], -1).max(axis=1) > 0)
    print("  max |difference| %.3e; planes that differ: %s ... (cuts at %s)" % (d.max(), zz[:12].tolist(), [z0 for z0, _ in parts]))
ref.close()
