"""dev: c5 for a few thousand steps -- finite fields, bounded mass drift, interface still sharp"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np
import bench
from openlbmpm_amd.rk3d import RK3DSlab
n = int(sys.argv[1]) if len(sys.argv) > 1 else 512
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 3000
dom = bench.c5_domain((n, n, n)); rR, rB = bench.c5_densities(dom, 0, n)
m0R, m0B = float(rR.sum()), float(rB.sum())
s = RK3DSlab(dom, 0, n, dict(relax=os.environ.get("LBMPM_K3_RELAX", "MRT"))); s.set_density(rR, rB)
t0 = time.perf_counter(); s.step_single(steps); s.sync(); dt = time.perf_counter() - t0
s.phase_field(diagnostics=True)
r, b, phi, vz = s.get("rhoR"), s.get("rhoB"), s.get("phi"), s.get("vz")
info = s.storage_info(); s.close()
print("%d^3, %d steps in %.1f s (%.0f MLUPS): finite %s; red mass x %.6f, blue mass x %.6f; max |u_z| %.3e; cells in single-colour rows %.1f %%" %
      (n, steps, dt, dom.sum() * steps / dt / 1e6, bool(np.isfinite(r).all() and np.isfinite(b).all() and np.isfinite(vz).all()),
       r.sum() / m0R, b.sum() / m0B, np.abs(vz).max(), 100.0 * info["cells_in_flagged_rows"] / info["fluid_cells"]))
