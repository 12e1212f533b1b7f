#!/usr/bin/env python3
"""Long run of the bench's c5 workload: the blue fluid enters through the Zou-He velocity plane only, so its mass
must grow by rho_in * |velocityZB| * (fluid cells of the inlet plane) per step; red leaves through the density outlet.
    python tools/soak_c5.py [n=512] [steps=3000] [every=500]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import bench
from openlbmpm_amd.rk3d import RK3DSlab

n = int(sys.argv[1]) if len(sys.argv) > 1 else 512
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 3000
every = int(sys.argv[3]) if len(sys.argv) > 3 else 500
dom = bench.c5_domain((n, n, n))
rR, rB = bench.c5_densities(dom, 0, n)
par = dict(relax=os.environ.get("LBMPM_K3_RELAX", "MRT"))
if os.environ.get("SOAK_RED_OUTLET"):
    # the shipped ini imposes BLUE at the outlet (densityBL = 1.0, densityRL = 1e-8, RKtwophasesetup3D.ini:37-38) although the
    # lattice starts red below the top buffer: blue then also enters from below.  With the resident fluid at the outlet
    # the blue mass has one source only, the velocity plane.
    par.update(densityRL=1.0, densityBL=1.0e-8)
s = RK3DSlab(dom, 0, n, par)
s.set_density(rR, rB)
inlet_cells = int(dom[n - 2].sum())            # the Zou-He plane (second from the top; the top plane is its ghost)
v = 1.0e-4
t0 = time.time(); done = 0; hist = []
while done <= steps:
    s.phase_field(diagnostics=True)
    R, B = s.get("rhoR"), s.get("rhoB")
    assert np.isfinite(R).all() and np.isfinite(B).all(), "non-finite at step %d" % done
    hist.append((done, float(R.sum()), float(B.sum())))
    z_front = int(np.argmax((B - R).sum(axis=(1, 2)) > 0))       # lowest plane where blue outweighs red
    print("step %5d  massR %.6e  massB %.6e  blue gained %.4e  expected %.4e  front plane %d  (%.0f s)"
          % (done, hist[-1][1], hist[-1][2], hist[-1][2] - hist[0][2], v * inlet_cells * done, z_front, time.time() - t0), flush=True)
    if done == steps:
        break
    m = min(every, steps - done)
    s.step_single(m); done += m
gain = hist[-1][2] - hist[0][2]
print("blue mass balance: gained / (rho v A t) = %.4f" % (gain / (v * inlet_cells * steps)))
