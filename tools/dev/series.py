"""dev: ms per step of the c5 bench kernel in consecutive windows of 10 steps (does the rate hold?):  python tools/dev/series.py [n=512] [windows=30] [state]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import bench
from openlbmpm_amd.rk3d import RK3DSlab
n = int(sys.argv[1]) if len(sys.argv) > 1 else 512
w = int(sys.argv[2]) if len(sys.argv) > 2 else 30
state = sys.argv[3] if len(sys.argv) > 3 else "initial"
dom = bench.c5_domain((n, n, n))
rR, rB = bench.c5_state(dom, 0, n, state)
s = RK3DSlab(dom, 0, n, dict(relax="MRT"))
s.set_density(rR, rB)
s.step_single(2); s.sync()
print(os.path.basename(os.environ.get("LBMPM_LIBRARY", "product")), state, " ".join("%.3f" % (s.step_timed(10)[0] / 10) for _ in range(w)), flush=True)
