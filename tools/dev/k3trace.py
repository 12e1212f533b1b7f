"""dev: when do the workgroups of one rk3dq_fused launch start, finish their prologue and end?  (LBMPM_RK3D_TRACE)"""
import sys, os, ctypes as C
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import devlib  # noqa: F401  -- the -DLBMPM_DEV build: the product library has no time stamps / knock-outs
os.environ["LBMPM_RK3D_TRACE"] = "1"
import numpy as np
from openlbmpm_amd.rk3d import RK3DSlab
from openlbmpm_amd.geometry import porous_spheres
import bench
nz = int(sys.argv[1]) if len(sys.argv) > 1 else 64
full = porous_spheres(512, 512, 512, seed=bench.SEED)
dom = np.ascontiguousarray(full[:nz]); dom[-10:] = full[-10:]
rR, rB = bench.c5_densities(dom, 0, nz)
s = RK3DSlab(dom, 0, nz, dict(relax="MRT"))
s.set_density(rR, rB)
s.step_single(5); s.sync()
ms = s.step_timed(10)[0] / 10
chunks = (nz + 63) // 64
nb = 8 * 8 * 8 * chunks
out = np.zeros(4 * nb, dtype=np.uint64)
s._L.lbmpm_rk3d_debug_trace(s._h, out.ctypes.data_as(C.c_void_p), nb)
t = out.reshape(nb, 4).astype(np.float64)
ok = t[:, 2] > 0
t0 = t[ok, 0].min()
st, pr, en = (t[ok, 0] - t0) / 100.0, (t[ok, 1] - t0) / 100.0, (t[ok, 2] - t0) / 100.0     # microseconds
print("nz %d: %.3f ms per step; %d workgroups traced" % (nz, ms, ok.sum()))
print("start   : min %.1f  median %.1f  max %.1f us" % (st.min(), np.median(st), st.max()))
print("prologue: median %.1f us (max %.1f)" % (np.median(pr - st), (pr - st).max()))
print("run     : median %.1f us  min %.1f  max %.1f" % (np.median(en - st), (en - st).min(), (en - st).max()))
print("end     : min %.1f  median %.1f  max %.1f us  -> kernel span %.1f us" % (en.min(), np.median(en), en.max(), en.max()))
first = st < np.median(st) - 1.0
print("first round: %d blocks, run median %.1f;  later: %d blocks, run median %.1f" % (first.sum(), np.median((en - st)[first]), (~first).sum(), np.median((en - st)[~first])))
nplanes = (t[ok, 3].astype(np.uint64) & np.uint64(0xffffffff)).astype(np.int64) - (t[ok, 3].astype(np.uint64) >> np.uint64(32)).astype(np.int64) + 1
print("per march step (run - prologue) / (planes + 2): median %.2f us" % np.median((en - pr) / (nplanes + 2)))

za = (out.reshape(nb, 4)[ok, 3] >> np.uint64(32)).astype(np.int64)
s2 = RK3DSlab  # noqa
for z0 in sorted(set(za.tolist())):
    m = za == z0
    print("  chunk from plane %3d: %4d blocks, per march step median %.2f us, start median %.0f us" % (z0, m.sum(), np.median(((en - pr) / (nplanes + 2))[m]), np.median(st[m])))
s.close()
