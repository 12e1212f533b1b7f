"""dev: single-domain step time of the bench lattice against the chunk length of rk3dq_fused (LBMPM_RK3D_CHUNK)"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import bench
from openlbmpm_amd.rk3d import RK3DSlab
n = int(sys.argv[1]) if len(sys.argv) > 1 else 512
dom = bench.c5_domain((n, n, n)); rR, rB = bench.c5_densities(dom, 0, n)
for ch in [int(v) for v in sys.argv[2:]] or [4, 8, 16, 32, 64, 128]:
    os.environ["LBMPM_RK3D_CHUNK"] = str(ch)
    s = RK3DSlab(dom, 0, n, dict(relax=os.environ.get("LBMPM_K3_RELAX", "MRT"))); s.set_density(rR, rB); s.step_single(3)
    t = min(s.step_timed(10)[0] / 10 for _ in range(3))
    print("chunk %3d: %.3f ms per step  (%d chunks, %.2f us per plane)" % (ch, t, (n + ch - 1) // ch, t * 1e3 / n), flush=True)
    s.close()
