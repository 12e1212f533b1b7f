"""dev check: q23 storage vs oracle and vs the 38-value kernels on small compact-storage cases"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), "tests"))
from helpers import rel_err
from openlbmpm_amd.geometry import porous_spheres, initial_densities_rk3d
from openlbmpm_amd.rk3d import RK3DCluster
from oracle.rk3d import RK3DOracle

def case(nx, ny, nz, seed):
    dom = porous_spheres(nx, ny, nz, porosity=0.7, rmin=3.0, rmax=7.0, seed=seed, nbuf=5)
    rR, rB = initial_densities_rk3d(dom, 5)
    return dom, rR, rB

F = ("rhoR", "rhoB", "phi", "vx", "vy", "vz")
for relax in ("SRT", "MRT"):
    for (nx, ny, nz, seed) in ((128, 21, 30, 5), (64, 19, 41, 12), (192, 40, 70, 3)):
        dom, rR, rB = case(nx, ny, nz, seed)
        par = dict(tauR=1.0, tauB=0.8, relax=relax)
        os.environ.pop("LBMPM_RK3D_STORAGE", None)
        c = RK3DCluster(dom, 1, par); c.set_density(rR, rB)
        print(relax, nx, ny, nz, c.slabs[0].dominant_kernel, flush=True)
        os.environ["LBMPM_RK3D_STORAGE"] = "38"
        d = RK3DCluster(dom, 1, par); d.set_density(rR, rB)
        os.environ.pop("LBMPM_RK3D_STORAGE", None)
        o = RK3DOracle(dom, rR, rB, par)
        for n in (1, 14, 25):
            c.step(n); d.step(n); o.run(n)
            c.observe(); d.observe(); o.macro()
            umax = max(float(np.max(np.abs(o.field(f)))) for f in ("vx", "vy", "vz"))
            eo = {f: rel_err(c.get(f), o.field(f), scale=umax if f[0] == "v" else None) for f in F}
            ed = {f: rel_err(c.get(f), d.get(f), scale=umax if f[0] == "v" else None) for f in F}
            print("  steps %3d  vs oracle %.2e   vs q38 %.2e   (q38 vs oracle %.2e)" % (
                c.slabs[0].steps_done, max(eo.values()), max(ed.values()),
                max(rel_err(d.get(f), o.field(f), scale=umax if f[0] == "v" else None) for f in F)), flush=True)
        c.close(); d.close()
