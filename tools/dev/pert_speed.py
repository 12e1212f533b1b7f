"""dev: the fused perturbation step (rk2dp_fused) at the c2 lattice size, against the kernel-by-kernel loop"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np
from openlbmpm_amd.rk2d import RK2DSolver
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
dom = np.ones((n, n), dtype=np.uint8); dom[:, 0] = 0; dom[:, -1] = 0
dom[2:-2, 1] = 1; dom[2:-2, -2] = 1
dom[:, 0] = 0; dom[:, -1] = 0
dom[[0, 1, -2, -1], :] = 1
W = np.array([4. / 9.] + [1. / 9.] * 4 + [1. / 36.] * 4)
rR = np.where(np.arange(n)[:, None] > 3 * n // 4, 1.0, 5e-8) * np.ones((n, n)); rB = np.where(np.arange(n)[:, None] > 3 * n // 4, 5e-8, 1.0) * np.ones((n, n))
for relax in ("SRT", "MRT"):
    s = RK2DSolver(dom, dict(relax=relax, inlet="Neumann", outlet="Dirichlet", vyR=-1e-4, vyB=0.0, rhoRL=5e-8, rhoBL=1.0, tauR=1.0, tauB=1.0, beta=0.7),
                   perturbation=dict(AkR=0.007, AkB=0.009, solidPhi=0.5))
    s.set_pdf(rR[..., None] * W * (dom == 1)[..., None], rB[..., None] * W * (dom == 1)[..., None])
    s.step(200)
    ms, dom_ms = s.step_timed(2000)
    nf = s.num_fluid_nodes
    print("rk2dp_fused %s %dx%d: %.4f ms per step (kernel %.4f), %.0f MLUPS, %.2f of 8 TB/s by 288 B per update" %
          (relax, n, n, ms / 2000, dom_ms / 2000, nf / (ms / 2000) / 1e3, 288.0 * nf / (dom_ms / 2000 * 1e-3) / 8e12), flush=True)
    assert np.isfinite(s.get("rec_rhoR")).all()
    s.close()
