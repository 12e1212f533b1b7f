"""dev: c5 kernel with both colours present in every cell (no single-colour shortcut anywhere) vs the bench's initial state"""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np
from openlbmpm_amd.rk3d import RK3DSlab
from openlbmpm_amd.geometry import porous_spheres
import bench

n = int(sys.argv[1]) if len(sys.argv) > 1 else 512
dom = porous_spheres(n, n, n, seed=bench.SEED)
for label, mix in (("bench initial state", None), ("two colours in every cell", 0.5), ("blue blob: 30 % of the planes mixed", -1)):
    rR, rB = bench.c5_densities(dom, 0, n)
    if mix is not None and mix > 0:
        f = (dom == 1)
        rR = np.where(f, 0.5, 0.0); rB = np.where(f, 0.5, 0.0)
    elif mix == -1:
        z = np.arange(n)[:, None, None]
        w = np.clip((z - 0.35 * n) / (0.3 * n), 0.0, 1.0) * np.ones(dom.shape)
        f = (dom == 1)
        rR = np.where(f, 1.0 - w, 0.0); rB = np.where(f, w, 0.0)
    for relax in ("MRT", "SRT"):
        s = RK3DSlab(dom, 0, n, dict(relax=relax))
        s.set_density(rR, rB)
        s.step_single(3); s.sync()
        ms_total, ms_dom = s.step_timed(20)
        nf = s.num_fluid_nodes
        print("%-40s %s  step %.3f ms  MLUPS %.0f" % (label, relax, ms_total / 20, nf * 20 / ms_total / 1e3), flush=True)
        s.close()
