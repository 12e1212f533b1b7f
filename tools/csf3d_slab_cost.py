"""What the slab decomposition of the 3-D CSF model costs in GPU work: the bench's porous lattice cut into k slabs that all run on THIS
GPU (rk3dcsf.RK3DCSFCluster: one context and stream pair per slab, face messages as device copies), against the undivided lattice.

    python tools/csf3d_slab_cost.py [edge=512] [steps=30] [relax=MRT] [slabs=1,2,4,8]

Prints per k: ms per step of the whole lattice, the share of the fluid cells on the bulk path, the bytes of the three face messages per
step and rank.  On a node every rank runs one slab: its step is ~ 1/k of the figure here plus the latency of three messages."""
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    edge = int(sys.argv[1]) if len(sys.argv) > 1 else 512
    steps = int(sys.argv[2]) if len(sys.argv) > 2 else 30
    relax = sys.argv[3] if len(sys.argv) > 3 else "MRT"
    ks = [int(v) for v in (sys.argv[4] if len(sys.argv) > 4 else "1,2,4,8").split(",")]
    from openlbmpm_amd.geometry import porous_spheres, initial_densities_rk3d
    from openlbmpm_amd.rk3dcsf import RK3DCSFCluster, MSG_PDF, MSG_PHI, MSG_NORMAL
    dom = porous_spheres(edge, edge, edge, porosity=0.65, rmin=6.0, rmax=20.0, seed=20260928, nbuf=10)
    dom[0] = dom[1]; dom[-1] = dom[-2]
    rR, rB = initial_densities_rk3d(dom, 10)
    par = dict(relax=relax, theta=60.0, tauB=0.8)
    ref = None
    for k in ks:
        c = RK3DCSFCluster(dom, par, nslabs=k)
        c.set_macro(rR, rB)
        c.step(10); c.sync()
        t0 = time.perf_counter()
        c.step(steps); c.sync()
        ms = (time.perf_counter() - t0) * 1e3 / steps
        msg = 0 if k == 1 else 8 * sum(c.slabs[0].face_doubles(m, f) for m in (MSG_PDF, MSG_PHI, MSG_NORMAL) for f in (0, 1))
        phi = c.get("phi")
        same = None if ref is None else bool(np.array_equal(phi, ref))
        ref = phi if ref is None else ref
        print(json.dumps(dict(workload="3-D CSF %s, %d^3 porous, initial state, steps 11..%d" % (relax, edge, 10 + steps), slabs=k, ms_per_step=round(ms, 3),
                              mlups=round(c.num_fluid_nodes / ms / 1e3, 1), bulk_share=round(c.bulk_cells / c.num_fluid_nodes, 4),
                              message_bytes_per_step_of_slab_0=msg, phi_equals_the_undivided_lattice=same)), flush=True)
        c.close()


if __name__ == "__main__":
    main()
