"""dev: workgroup time stamps of the INTERIOR launch of one slab rank (the last rk3dq_fused launch of lbmpm_rk3d_step_slab)"""
import sys, os, ctypes as C
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import devlib  # noqa: F401  -- the -DLBMPM_DEV build: the product library has no time stamps / knock-outs
os.environ["LBMPM_RK3D_TRACE"] = "1"
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
mode = sys.argv[2] if len(sys.argv) > 2 else "slab"
with torch.cuda.stream(st):
    if mode == "slab":
        s.step_slab(5, True, True, exchange); s.step_slab(10, True, True, exchange, timed=(os.environ.get("K3_TIMED", "1") == "1"))
        print(s.slab_timing())
    elif mode == "ib":    # interior on the second stream, then the boundary planes: no pack / exchange / unpack at all
        for _ in range(15):
            s.collide_interior(); s.collide_boundary()
    else:           # the same planes as ONE launch on the context's stream, no exchange (results wrong at the faces, timing only)
        for _ in range(15):
            s.collide()
s.sync()
nb = 8 * 8 * 8 * (2 if mode == 'single' else 1)
out = np.zeros(4 * nb, dtype=np.uint64)
s._L.lbmpm_rk3d_debug_trace(s._h, out.ctypes.data_as(C.c_void_p), nb)
t = out.reshape(nb, 4).astype(np.float64); ok = t[:, 2] > 0
t0 = t[ok, 0].min(); stt, pr, en = (t[ok, 0] - t0) / 100.0, (t[ok, 1] - t0) / 100.0, (t[ok, 2] - t0) / 100.0
w = out.reshape(nb, 4)[ok, 3]
npl = (w & np.uint64(0xffffffff)).astype(np.int64) - (w >> np.uint64(32)).astype(np.int64) + 1
print(mode, "rank", r, "planes", parts[r][1], ": blocks", ok.sum(), "planes per block", int(np.median(npl)), " kernel span %.0f us" % en.max(),
      " per march step median %.2f us" % np.median((en - pr) / (npl + 2)), " start median %.0f max %.0f" % (np.median(stt), stt.max()))
za = (w >> np.uint64(32)).astype(np.int64)
for z0 in sorted(set(za.tolist())):
    m = za == z0
    print("  blocks from plane %3d: %4d, planes %d, start %.0f .. %.0f us, run median %.1f us (prologue %.1f), end max %.0f" % (
        z0, m.sum(), int(np.median(npl[m])), stt[m].min(), stt[m].max(), np.median((en - stt)[m]), np.median((pr - stt)[m]), en[m].max()))
bids = np.arange(nb)[ok]
print("  run median by bid & 7:", " ".join("%.0f" % np.median((en - stt)[(bids & 7) == k]) for k in range(8)),
      "| by round: first %.0f later %.0f" % (np.median((en - stt)[stt < 50]), np.median((en - stt)[stt >= 50])))
for q in slabs.values():
    q.close()
