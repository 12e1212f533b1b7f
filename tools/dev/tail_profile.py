"""dev: how far does the minority colour reach?  per plane: max over fluid cells of rho_B / rho (log10) below the front, at a few times"""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np
from openlbmpm_amd.rk3d import RK3DCluster
from openlbmpm_amd.geometry import porous_spheres, initial_densities_rk3d
vz = float(sys.argv[1]) if len(sys.argv) > 1 else -2e-2
allfluid = len(sys.argv) > 2 and sys.argv[2] == "fluid"
dom = porous_spheres(128, 24, 96, porosity=0.7, rmin=3.0, rmax=7.0, seed=31, nbuf=5)
if allfluid:
    dom[:] = 1; dom[:, :, 0] = 0; dom[:, :, -1] = 0
rR, rB = initial_densities_rk3d(dom, 5)
c = RK3DCluster(dom, 1, dict(relax="MRT", velocityZB=vz))
c.set_density(rR, rB)
done = 0
for t in (50, 200, 500, 1000):
    c.step(t - done); done = t
    c.observe()
    r, b = c.get("rhoR"), c.get("rhoB")
    fl = dom == 1
    frac = np.where(fl, b / np.where(fl, r + b, 1.0), 0.0)
    mx = frac.reshape(96, -1).max(axis=1)
    print("step %4d  log10 max rhoB/rho per plane (z = 95 .. 0, every 3rd):" % t,
          " ".join("%4.0f" % (np.log10(v) if v > 0 else -999) for v in mx[::-3]), flush=True)
    print("           storage:", c.slabs[0].storage_info())
