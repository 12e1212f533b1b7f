"""Times the 3-D CSF colour-gradient step (lbmpm_rk3dcsf_*, four launches per step) on the bench's porous geometry.

    python tools/csf3d_bench.py [edge=256] [steps=20] [relax=MRT] [state=initial|mixed]

Prints ms per step, MLUPS on fluid cells, and the share of the dominant kernel (csf3d_collide) by HIP events."""
import json
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    edge = int(sys.argv[1]) if len(sys.argv) > 1 else 256
    steps = int(sys.argv[2]) if len(sys.argv) > 2 else 20
    relax = sys.argv[3] if len(sys.argv) > 3 else "MRT"
    state = sys.argv[4] if len(sys.argv) > 4 else "initial"
    from openlbmpm_amd.geometry import porous_spheres, initial_densities_rk3d
    from openlbmpm_amd.rk3dcsf import RK3DCSFSolver
    dom = porous_spheres(edge, edge, edge, porosity=0.65, rmin=6.0, rmax=20.0, seed=20260928, nbuf=10)
    dom[0] = dom[1]; dom[-1] = dom[-2]
    rR, rB = initial_densities_rk3d(dom, 10)
    if state == "mixed":
        fl = dom == 1
        rR = np.where(fl, 0.5, 0.0); rB = np.where(fl, 0.5, 0.0)
        zz = np.arange(edge)[:, None, None]
        rR = rR * (1.0 + 0.2 * np.sin(zz / 7.0)); rB = rB * (1.0 - 0.2 * np.sin(zz / 7.0))
    s = RK3DCSFSolver(dom, dict(relax=relax, theta=60.0, tauB=0.8))
    s.set_macro(rR, rB)
    s.step(3); s.sync()
    tot, dom_ms = s.step_timed(steps)
    n = s.num_fluid_nodes
    print(json.dumps(dict(workload="3-D CSF colour gradient %s, %d^3 porous (porosity 0.65), state %s" % (relax, edge, state), fluid_cells=n,
                          wetting_solids=s.num_wetting_solids, ms_per_step=tot / steps, mlups=n * steps / tot / 1e3, collide_ms=dom_ms / steps,
                          device_gb=s.device_bytes / 1e9, finite=bool(np.isfinite(s.get("rhoR")).all()))))


if __name__ == "__main__":
    main()
