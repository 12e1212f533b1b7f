"""dev: where a stored row flag says 'single colour' but the observed density of the other colour is not zero"""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np
from openlbmpm_amd.rk3d import RK3DCluster
from openlbmpm_amd.geometry import porous_spheres, initial_densities_rk3d
dom = porous_spheres(128, 24, 192, porosity=0.7, rmin=3.0, rmax=7.0, seed=31, nbuf=5)
rR, rB = initial_densities_rk3d(dom, 5)
c = RK3DCluster(dom, 1, dict(relax="MRT", velocityZB=-2e-2))
c.set_density(rR, rB)
nz, ny, nx = dom.shape
fluid = dom == 1
s = c.slabs[0]
c.step(150)
bad = 0
for t in range(150, 230):
    c.observe()
    r, b = c.get("rhoR"), c.get("rhoB")
    c.step(1)
    only_red = np.where(fluid, np.abs(b) <= 2.0 ** -51 * (r + b), True).reshape(nz, ny, nx // 64, 64).all(axis=3)
    only_blue = np.where(fluid, np.abs(r) <= 2.0 ** -51 * (r + b), True).reshape(nz, ny, nx // 64, 64).all(axis=3)
    want = only_red.astype(np.int64) + 2 * only_blue.astype(np.int64)
    has = fluid.reshape(nz, ny, nx // 64, 64).any(axis=3)
    for z in range(1, nz - 1):
        got = s.debug_plane(24, z + 1).astype(np.int64)
        d = (got != want[z]) & has[z]
        if d.any():
            for y, sg in zip(*np.nonzero(d)):
                seg = slice(sg * 64, sg * 64 + 64)
                bb = b[z, y, seg][fluid[z, y, seg]]; rr = r[z, y, seg][fluid[z, y, seg]]
                print("step %d plane %d row %d seg %d: flag %d want %d; rhoB nonzero: %s; min rhoR %.3g" % (t + 1, z, y, sg, got[y, sg], want[z][y, sg],
                      ["%.3g" % v for v in bb[bb != 0][:6]], rr.min()), flush=True)
                bad += 1
    if bad > 12:
        break
print("mismatches:", bad)
