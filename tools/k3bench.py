#!/usr/bin/env python3
"""3-D kernel timing aid: all-fluid vs porous, several sizes (tuning only)."""
import sys, os, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from openlbmpm_amd.rk3d import RK3DSlab
from openlbmpm_amd.geometry import porous_spheres
import bench

def run(dom, label, steps=20):
    nz = dom.shape[0]
    rR, rB = bench.c5_densities(dom, 0, nz)
    s = RK3DSlab(dom, 0, nz, dict(relax=os.environ.get("LBMPM_K3_RELAX", "SRT")))
    s.set_density(rR, rB)
    s.step_single(3); s.sync()
    ms_total, ms_dom = s.step_timed(steps)
    nf = s.num_fluid_nodes
    print("%-28s fluid %.1fM  step %.3f ms  collide %.3f ms  MLUPS %.0f  collide GB/s(alg) %.0f" % (
        label, nf / 1e6, ms_total / steps, ms_dom / steps, nf * steps / ms_total / 1e3,
        608.0 * nf / (ms_dom / steps * 1e-3) / 1e9), flush=True)
    s.close()

n = int(sys.argv[1]) if len(sys.argv) > 1 else 256
which = sys.argv[2] if len(sys.argv) > 2 else "both"
envs = sys.argv[3:] or [""]          # e.g. LBMPM_RK3D_TILE=1,LBMPM_RK3D_CHUNK=32  (one run per argument)
doms = []
if which in ("both", "fluid"):
    doms.append((np.ones((n, n, n), dtype=np.uint8), "all fluid %d^3" % n))
if which in ("both", "porous"):
    doms.append((porous_spheres(n, n, n, seed=bench.SEED), "porous %d^3" % n))
for e in envs:
    for kv in filter(None, e.split(",")):
        k, v = kv.split("=")
        os.environ[k] = v
        if k in ("LBMPM_RK3D_TILE", "LBMPM_RK3D_CHUNK", "LBMPM_RK3D_FILL", "LBMPM_RK3D_BOUNDARY", "LBMPM_RK3D_XCC"):
            # knobs of the development build: contexts from here on come from that library
            from openlbmpm_amd import _lib, build
            _lib.use_library(build.build_dev_if_stale(verbose=False))
    for dom, label in doms:
        run(dom, (label + " " + e).strip())
