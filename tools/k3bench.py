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
    s = RK3DSlab(dom, 0, nz)
    s.set_density(rR, rB)
    s.step_single(3); s.sync()
    ms_total, ms_dom = s.step_timed(steps)
    nf = s.num_fluid_nodes
    print("%-28s fluid %.1fM  step %.3f ms  collide %.3f ms  MLUPS %.0f  collide GB/s(alg) %.0f" % (
        label, nf / 1e6, ms_total / steps, ms_dom / steps, nf * steps / ms_total / 1e3,
        608.0 * nf / (ms_dom / steps * 1e-3) / 1e9), flush=True)
    s.close()

n = int(sys.argv[1]) if len(sys.argv) > 1 else 256
dom = np.ones((n, n, n), dtype=np.uint8)
run(dom, "all fluid %d^3" % n)
run(porous_spheres(n, n, n, seed=bench.SEED), "porous %d^3" % n)
