"""dev: how does the marching kernel's time per plane depend on the number of planes (launch size) and the chunk length?"""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np
from openlbmpm_amd.rk3d import RK3DSlab
from openlbmpm_amd.geometry import porous_spheres
import bench
full = porous_spheres(512, 512, 512, seed=bench.SEED)
for nz in (512, 128, 64, 32):
    for chunk in ("32", "16", "64"):
        os.environ["LBMPM_RK3D_CHUNK"] = chunk
        dom = np.ascontiguousarray(full[:nz]); dom[-10:] = full[-10:]
        rR, rB = bench.c5_densities(dom, 0, nz)
        s = RK3DSlab(dom, 0, nz, dict(relax="MRT"))
        s.set_density(rR, rB)
        s.step_single(3); s.sync()
        ms = min(s.step_timed(20)[0] / 20 for _ in range(3))
        nf = s.num_fluid_nodes
        print("nz %3d chunk %s: %.3f ms/step  %.4f ms per plane  MLUPS %.0f" % (nz, chunk, ms, ms / nz, nf / ms / 1e3), flush=True)
        s.close()
