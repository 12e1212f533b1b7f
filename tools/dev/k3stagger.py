import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import devlib  # noqa: F401  -- the -DLBMPM_DEV build: the product library has no time stamps / knock-outs
import numpy as np
from openlbmpm_amd.rk3d import RK3DSlab
from openlbmpm_amd.geometry import porous_spheres
import bench
full = porous_spheres(512, 512, 512, seed=bench.SEED)
for nz in (64, 512):
    for dbg in ("0", "1", "2", "4"):
        os.environ["LBMPM_RK3D_DBG"] = dbg
        dom = np.ascontiguousarray(full[:nz]); dom[-10:] = full[-10:]
        rR, rB = bench.c5_densities(dom, 0, nz)
        s = RK3DSlab(dom, 0, nz, dict(relax="MRT"))
        s.set_density(rR, rB)
        s.step_single(3); s.sync()
        ms = min(s.step_timed(20)[0] / 20 for _ in range(3))
        print("nz %3d stagger %s: %.3f ms/step" % (nz, dbg, ms), flush=True)
        s.close()
