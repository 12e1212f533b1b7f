"""dev: do the kernels that use scratch memory give the same bits twice on a big lattice?  (Round 3: two face kernels with 320 B of
scratch per lane lost the stores of whole waves at random at >= 600 waves.)  Here: the two-sweep iso-8 / iso-10 Shan-Chen step
(sc2d_iso_collide, 304 B per lane) and the set-up kernels (sc2d_init_*), 2048^2 porous, two runs of each."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np
from openlbmpm_amd.sc2d import SC2DSolver
from openlbmpm_amd.geometry import porous_disks, image_domain
n = int(sys.argv[1]) if len(sys.argv) > 1 else 2048
img = porous_disks(n, n - 40, porosity=0.65, rmin=6.0, rmax=20.0, seed=7)
dom = image_domain(img, 20, 0.5)
ny = dom.shape[0]
lower = (np.arange(ny)[:, None] + np.zeros(dom.shape, dtype=np.int64)) < ny - 20
fluid = dom == 1
r0 = np.where(fluid & lower, 1.0, 0.0) + np.where(fluid & ~lower, 0.02, 0.0)
r1 = np.where(fluid & lower, 0.02, 0.0) + np.where(fluid & ~lower, 1.0, 0.0)
for scheme, relax in ((8, "SRT"), (10, "SRT"), (8, "MRT"), (4, "MRT")):
    out = []
    for rep in range(3):
        s = SC2DSolver(dom, dict(inter="EFS", relax=relax, outlet="Dirichlet", scheme=scheme), diagnostics=True)
        s.set_density(r0, r1)
        s.step(12)
        out.append([s.get(f) for f in ("rho0", "rho1", "f0", "f1")])
        s.close()
    same = all(np.array_equal(a, b) for rep in out[1:] for a, b in zip(out[0], rep))
    fin = all(np.isfinite(a).all() for a in out[0])
    print("scheme %2d %s %dx%d: three runs bit-equal: %s, finite: %s" % (scheme, relax, n, ny, same, fin), flush=True)
