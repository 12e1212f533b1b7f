"""per-rank GPU work of the 3-D CSF slabs: a 512 x 512 x (64 k) lattice in k slabs of 64 planes (what a rank of an 8-rank 512^3 run holds),
k = 1 (undivided, no ghosts) against k = 2: python tools/dev/csf_slab_rank.py"""
import os, sys, time, json
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from openlbmpm_amd.geometry import porous_spheres
from openlbmpm_amd.rk3dcsf import RK3DCSFCluster

nz = 132
dom = porous_spheres(512, 512, nz, porosity=0.65, rmin=6.0, rmax=20.0, seed=20260928, nbuf=10)
dom[0] = dom[1]; dom[-1] = dom[-2]
fl = dom == 1
for state in ([os.environ["CSF_TL_ONLY"]] if os.environ.get("CSF_TL_ONLY") else ["bulk", "mixed"]):
    if state == "bulk":
        zz = np.arange(nz)[:, None, None]
        rR, rB = np.where(fl & (zz < nz - 10), 1.0, 0.0), np.where(fl & (zz >= nz - 10), 1.0, 0.0)
    else:
        rR, rB = np.where(fl, 0.5, 0.0), np.where(fl, 0.5, 0.0)
    for k, cuts in (((2, [0, nz - 4, nz]),) if os.environ.get("CSF_TL_ONLY") else ((1, None), (2, [0, 66, nz]), (2, [0, nz - 4, nz]))):
        c = RK3DCSFCluster(dom, dict(relax="MRT", theta=60.0, tauB=0.8), nslabs=k, cuts=cuts)
        c.set_macro(rR, rB)
        c.step(10); c.sync()
        t0 = time.perf_counter(); c.step(30); c.sync()
        ms = (time.perf_counter() - t0) * 1e3 / 30
        print(json.dumps(dict(state=state, slabs=k, cuts=cuts, ms_per_step=round(ms, 3), ms_per_slab=round(ms / k, 3),
                              bulk_share=round(c.bulk_cells / c.num_fluid_nodes, 3))), flush=True)
        c.close()
