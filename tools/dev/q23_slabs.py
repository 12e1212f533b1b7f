"""dev: q23 k-slab cluster == single domain, bit for bit"""
import os, sys
import numpy as np
R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "tests"))
from openlbmpm_amd.geometry import porous_spheres, initial_densities_rk3d
from openlbmpm_amd.rk3d import RK3DCluster
F = ("rhoR", "rhoB", "phi", "vx", "vy", "vz")
for (nx, ny, nz, seed) in ((64, 19, 41, 12), (128, 21, 38, 5)):
    dom = porous_spheres(nx, ny, nz, porosity=0.7, rmin=3.0, rmax=7.0, seed=seed, nbuf=5)
    rR, rB = initial_densities_rk3d(dom, 5)
    for relax in ("SRT", "MRT"):
        par = dict(relax=relax, tauR=0.9, tauB=0.7)
        out = {}
        for k in (1, 2, 3, 5):
            c = RK3DCluster(dom, k, par); c.set_density(rR, rB)
            assert c.slabs[0].dominant_kernel == "rk3dq_fused", c.slabs[0].dominant_kernel
            res = []
            for n in (1, 10):
                c.step(n); c.observe()
                res.append({f: c.get(f) for f in F})
            out[k] = res
            c.close()
            if k > 1:
                worst = max(float(np.max(np.abs(out[k][i][f] - out[1][i][f]))) for i in range(2) for f in F)
                eq = all(np.array_equal(out[k][i][f], out[1][i][f]) for i in range(2) for f in F)
                print(nx, ny, nz, relax, "k=%d" % k, "bitwise" if eq else "DIFFERS max abs %.3e" % worst, flush=True)
