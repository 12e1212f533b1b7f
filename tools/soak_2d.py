#!/usr/bin/env python3
"""Long runs of the bench's 2-D workloads (c2, c3, c4): finiteness and mass drift after thousands of steps.
    python tools/soak_2d.py [steps=20000]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import bench

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 20000
for name, build, size in (("c2", bench.build_c2, (1024, 1024)), ("c3", bench.build_c3, (2048, 2048)), ("c4", bench.build_c4, (2048, 2048))):
    s, m0, mass = build(size[0], size[1], 0)
    t0 = time.time()
    s.step(steps - 1); s.sync()
    dt = time.time() - t0
    if name == "c3":
        s.enable_diagnostics(True)                 # densities are recorded by the step that follows
    s.step(1)
    if name == "c3":
        r = s.get("rho0") + s.get("rho1")
    else:
        r = s.get("rhoR") + s.get("rhoB")
    ok = bool(np.isfinite(r).all())
    extra = ""
    if name == "c4":
        c = s.get_tracer(0); ok = ok and bool(np.isfinite(c).all()); extra = "  tracer min %.3g max %.3g" % (c.min(), c.max())
    print("%s: %d steps in %.1f s (%.0f MLUPS)  finite %s  mass drift %.3e  rho min %.4f max %.4f%s"
          % (name, steps, dt, s.num_fluid_nodes * steps / dt / 1e6, ok, float(r.sum()) / m0 - 1.0, r[r > 0].min(), r.max(), extra), flush=True)
    s.close()
