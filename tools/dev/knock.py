"""dev: timing of knock-out builds of rk3dq_fused (results wrong): LBMPM_LIBRARY=tools/dev/_build/kN.so python tools/dev/knock.py"""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np
from openlbmpm_amd.rk3d import RK3DSlab
import bench
n = 512
dom = bench.c5_domain((n, n, n))
for state in ("initial", "mixed"):
    rR, rB = bench.c5_state(dom, 0, n, state)
    s = RK3DSlab(dom, 0, n, dict(relax="MRT"))
    s.set_density(rR, rB)
    s.step_single(3); s.sync()
    ms, _ = s.step_timed(20)
    print("%-8s %s step %.3f ms" % (os.path.basename(os.environ.get("LBMPM_LIBRARY", "product")), state, ms / 20), flush=True)
    s.close()
