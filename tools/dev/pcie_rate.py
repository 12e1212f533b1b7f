"""dev: the c5 workload INCLUDING the host transfers of the boundary: set_density from host arrays, N steps, observe,
the five macroscopic fields back to the host; and the exact state out and in (get_state / set_state) on a 512 x 512 x 128 slab"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np
import bench
from openlbmpm_amd.rk3d import RK3DSlab
n = 512
dom = bench.c5_domain((n, n, n))
rR, rB = bench.c5_densities(dom, 0, n)
s = RK3DSlab(dom, 0, n, dict(relax="MRT"))
for steps in (100, 1000):
    t0 = time.perf_counter()
    s.set_density(rR, rB)
    t1 = time.perf_counter()
    s.step_single(steps); s.sync()
    t2 = time.perf_counter()
    s.phase_field(diagnostics=True)
    out = [s.get(f) for f in ("rhoR", "rhoB", "vx", "vy", "vz")]
    t3 = time.perf_counter()
    nf = s.num_fluid_nodes
    print("512^3, %4d steps: upload %.2f s, steps %.2f s (%.0f MLUPS), observe + 5 fields to the host %.2f s -> %.0f MLUPS including the transfers"
          % (steps, t1 - t0, t2 - t1, nf * steps / (t2 - t1) / 1e6, t3 - t2, nf * steps / (t3 - t0) / 1e6), flush=True)
s.close()
dom = bench.c5_domain((n, n, 128))
rR, rB = bench.c5_densities(dom, 0, 128)
s = RK3DSlab(dom, 0, 128, dict(relax="MRT"))
s.set_density(rR, rB); s.step_single(10)
t0 = time.perf_counter(); st, info = s.get_state(); t1 = time.perf_counter(); s.set_state(st, info["steps"], info["post_collision"]); t2 = time.perf_counter()
print("512 x 512 x 128 slab: get_state %.1f GB in %.2f s (%.1f GB/s), set_state %.2f s (%.1f GB/s)" % (st.nbytes / 1e9, t1 - t0, st.nbytes / 1e9 / (t1 - t0), t2 - t1, st.nbytes / 1e9 / (t2 - t1)))
