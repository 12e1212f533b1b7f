"""dev: wall-clock time of lbmpm_rk3d_step_slab for one rank, with and without the per-phase events"""
import sys, os, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np, torch
import bench
from openlbmpm_amd.rk3d import RK3DSlab, RK3DDistributed
K, r = 8, int(sys.argv[1]) if len(sys.argv) > 1 else 3
dom = bench.c5_domain((512, 512, 512))
rR, rB = bench.c5_densities(dom, 0, 512)
parts = RK3DDistributed.partition(dom, K)
st = torch.cuda.Stream(0)
slabs = {}
for q in (r - 1, r, r + 1):
    z0, nz = parts[q]
    s = RK3DSlab(dom, z0, nz, dict(relax="MRT")); s.set_density(rR[z0:z0 + nz], rB[z0:z0 + nz]); s.use_torch_stream(st); slabs[q] = s
with torch.cuda.stream(st):
    for s in slabs.values():
        s.pack()
s = slabs[r]
def exchange(what):
    s.buffer("f_recv_below").copy_(slabs[r - 1].buffer("f_send_up")); s.buffer("f_recv_above").copy_(slabs[r + 1].buffer("f_send_down"))
def noexchange(what):
    pass
for label, cb, timed in (("events on ", exchange, True), ("events off", exchange, False), ("no copies ", noexchange, False)):
    with torch.cuda.stream(st):
        s.step_slab(5, True, True, cb)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        s.step_slab(40, True, True, cb, timed=timed)
        torch.cuda.synchronize(); dt = time.perf_counter() - t0
    print("%s: %.3f ms per step (wall)" % (label, dt / 40 * 1e3), flush=True)
with torch.cuda.stream(st):
    for _ in range(5): s.collide()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(40): s.collide()
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
print("one launch of all planes: %.3f ms per step (wall)" % (dt / 40 * 1e3))
for q in slabs.values():
    q.close()
