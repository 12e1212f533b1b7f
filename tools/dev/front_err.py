"""dev: q23 vs 38-value kernels while a front sweeps a porous lattice: error growth"""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
from helpers import rel_err
from openlbmpm_amd.rk3d import RK3DCluster
from openlbmpm_amd.geometry import porous_spheres, initial_densities_rk3d
vz = float(sys.argv[1]); relax = sys.argv[2] if len(sys.argv) > 2 else "MRT"
dom = porous_spheres(128, 24, 192, porosity=0.7, rmin=3.0, rmax=7.0, seed=31, nbuf=5)
rR, rB = initial_densities_rk3d(dom, 5)
runs = {}
for st in ("23", "38"):
    os.environ["LBMPM_RK3D_STORAGE"] = st
    c = RK3DCluster(dom, 1, dict(relax=relax, velocityZB=vz)); c.set_density(rR, rB); runs[st] = c
done = 0
for t in (50, 100, 200, 400, 800, 1200, 1600, 2400, 3200):
    for c in runs.values():
        c.step(t - done); c.observe()
    done = t
    a, b = runs["23"], runs["38"]
    print("v %g %s step %4d:" % (vz, relax, t), " ".join("%s %.2e" % (f, rel_err(a.get(f), b.get(f))) for f in ("rhoR", "rhoB", "phi", "vz")),
          " max|u| %.3g" % np.abs(b.get("vz")).max(), flush=True)
