import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np
from openlbmpm_amd.rk3d import RK3DCluster
from openlbmpm_amd.geometry import initial_densities_rk3d
dom = np.ones((192, 24, 256), dtype=np.uint8); dom[:, :, 0] = 0; dom[:, :, -1] = 0
rR, rB = initial_densities_rk3d(dom, 5)
c = RK3DCluster(dom, 1, dict(relax="MRT", velocityZB=-2e-2, SolidRhoR=0.5, SolidRhoB=0.5)); c.set_density(rR, rB)
done = 0
for t in (200, 500, 800, 1200, 1600):
    c.step(t - done); done = t; c.observe()
    r, b = c.get("rhoR"), c.get("rhoB")
    fr = r / np.maximum(r + b, 1e-300)
    mid = fr[:, :, 64:192]
    print(t, "log10 max red fraction in the middle segments per plane (z=191..0 every 6th):", " ".join("%4.0f" % (np.log10(v) if v > 0 else -999) for v in mid.reshape(192, -1).max(axis=1)[::-6]))
    print(t, "log10 MIN red fraction:", " ".join("%4.0f" % (np.log10(v) if v > 0 else -999) for v in np.abs(mid).reshape(192, -1).min(axis=1)[::-6]))
