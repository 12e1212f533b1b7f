import sys, os
sys.path.insert(0, "tests"); sys.path.insert(0, ".")
import numpy as np
from test_rk3d_csf_gpu import _slab_case, solver
from openlbmpm_amd.rk3dcsf import RK3DCSFCluster
dom, rR, rB = _slab_case()
par = dict(relax="MRT", theta=55.0, tauB=0.8, velocityZR=0.0, velocityZB=-3.0e-3, sigma=0.06)
a = solver(dom, par); c = RK3DCSFCluster(dom, par, nslabs=2, diagnostics=True)
print("cuts", c.cuts, [s.ghost for s in c.slabs], [s.shape for s in c.slabs])
a.set_macro(rR, rB); c.set_macro(rR, rB)
for k in (1, 2):
    if k:
        a.step(1); c.step(1)
    for f in ("phi", "Gz", "K", "Fz", "rhoR", "fR", "fB", "rec_phi"):
        d = np.abs(a.get(f) - c.get(f))
        if d.ndim == 4:
            zs = np.nonzero(d.max(axis=(1, 2, 3)))[0]; qs = np.nonzero(d.max(axis=(0, 1, 2)))[0]
        else:
            zs = np.nonzero(d.max(axis=(1, 2)))[0]; qs = []
        print(k, f, float(d.max()), "planes", list(zs), "dirs", list(qs))
