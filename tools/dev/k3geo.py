"""dev: 3-D kernel timing on synthetic masks that separate 'idle lanes' from 'ragged rows'"""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np
from openlbmpm_amd.rk3d import RK3DSlab
import bench

def run(dom, label, steps=20):
    nz = dom.shape[0]
    rR, rB = bench.c5_densities(dom, 0, nz)
    s = RK3DSlab(dom, 0, nz, dict(relax=os.environ.get("LBMPM_K3_RELAX", "MRT")))
    s.set_density(rR, rB)
    s.step_single(3); s.sync()
    ms_total, ms_dom = s.step_timed(steps)
    nf = s.num_fluid_nodes
    print("%-34s fluid %.1fM (%.0f %%)  step %.3f ms  MLUPS %.0f  ns/cell %.4f" % (
        label, nf / 1e6, 100.0 * nf / dom.size, ms_total / steps, nf * steps / ms_total / 1e3, ms_total / steps * 1e6 / dom.size), flush=True)
    s.close()

n = int(sys.argv[1]) if len(sys.argv) > 1 else 512
base = np.ones((n, n, n), dtype=np.uint8)
d = base.copy(); d[:, ::3, :] = 0; d[:10] = 1; d[-10:] = 1
run(d, "every third row solid")
d = base.copy(); d[:, :, 40:64] = 0; d[:, :, 104:128] = 0; d[:10] = 1; d[-10:] = 1
run(d, "24 of every 64 columns solid (2 segs)")
x = np.arange(n)
d = base.copy(); d[:, :, (x % 64) >= 40] = 0; d[:10] = 1; d[-10:] = 1
run(d, "24 of every 64 columns solid")
rng = np.random.default_rng(1)
d = base.copy(); m = rng.random((n, n, n // 8)) < 0.35; d[np.repeat(m, 8, axis=2)] = 0; d[:10] = 1; d[-10:] = 1
run(d, "random 8-cell runs, 35 % solid")
