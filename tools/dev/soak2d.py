"""dev: the 2-D bench lattices over many steps on whatever library is loaded: finiteness, mass, density range"""
import sys; import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, bench, time
for name, build, size, steps in (("c2", bench.build_c2, 1024, 20000), ("c3", bench.build_c3, 2048, 6000), ("c4", bench.build_c4, 2048, 6000), ("c2p", bench.build_c2p, 1024, 20000)):
    s, m0, massf = build(size, size, 0)
    t=time.time(); s.step(steps); s.sync(); dt=time.time()-t
    if name in ("c2","c4","c2p"):
        s.enable_diagnostics(True); s.step(1)
        r = s.get("rhoR") + s.get("rhoB")
    else:
        s.enable_diagnostics(True); s.step(1)
        r = s.get("rho0") + s.get("rho1")
    print(name, "steps", steps, "%.1f s" % dt, "finite", bool(np.isfinite(r).all()), "mass %.6e -> %.6e (%.2e)" % (m0, r.sum(), r.sum()/m0-1), "rho range %.3f %.3f" % (r[r>0].min(), r.max()), flush=True)
    s.close()
