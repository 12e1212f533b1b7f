"""Long run of the 3-D CSF model on the bench's porous geometry: finiteness, colour masses against the inlet flux, density range, share of
the lattice on the bulk path, ms per step in windows.

    python tools/soak_csf3d.py [edge=256] [steps=3000] [relax=MRT] [window=500] [bulk_epsilon=0]
"""
import json
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    edge = int(sys.argv[1]) if len(sys.argv) > 1 else 256
    steps = int(sys.argv[2]) if len(sys.argv) > 2 else 3000
    relax = sys.argv[3] if len(sys.argv) > 3 else "MRT"
    window = int(sys.argv[4]) if len(sys.argv) > 4 else 500
    eps = float(sys.argv[5]) if len(sys.argv) > 5 else 0.0
    from openlbmpm_amd.geometry import porous_spheres, initial_densities_rk3d
    from openlbmpm_amd.rk3dcsf import RK3DCSFSolver
    dom = porous_spheres(edge, edge, edge, porosity=0.65, rmin=6.0, rmax=20.0, seed=20260928, nbuf=10)
    dom[0] = dom[1]; dom[-1] = dom[-2]
    rR, rB = initial_densities_rk3d(dom, 10)
    # a drainage: blue pushed in from the top at the 3-D ini's velocity, red leaves through the pressure outlet
    par = dict(relax=relax, theta=60.0, tauB=0.8, velocityZR=0.0, velocityZB=-1.0e-4, densityBL=1.0e-8, densityRL=1.0, bulk_epsilon=eps)
    s = RK3DCSFSolver(dom, par)
    s.set_macro(rR, rB)
    n = s.num_fluid_nodes
    m0 = (float(rR.sum()), float(rB.sum()))
    rows = []
    done = 0
    while done < steps:
        k = min(window, steps - done)
        tot, _ = s.step_timed(k)
        done += k
        a, b = s.get("rhoR"), s.get("rhoB")
        fl = dom == 1
        rows.append(dict(step=done, ms_per_step=round(tot / k, 4), mlups=round(n * k / tot / 1e3, 1), bulk_share=round(s.bulk_cells / n, 4),
                         finite=bool(np.isfinite(a).all() and np.isfinite(b).all()), massR=float(a.sum()) / m0[0], massB=float(b.sum()) / max(m0[1], 1e-300),
                         rho_min=float((a + b)[fl].min()), rho_max=float((a + b)[fl].max()), rhoR_min=float(a[fl].min()), rhoB_min=float(b[fl].min())))
        print(json.dumps(rows[-1]), flush=True)
    assert all(r["finite"] for r in rows)


if __name__ == "__main__":
    main()
