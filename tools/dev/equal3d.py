"""dev: a c5-model lattice stepped through whatever library LBMPM_LIBRARY names; densities and phase field written to argv[1]; with two file
arguments: compare bit for bit.   python tools/dev/equal3d.py /tmp/a.npz; LBMPM_LIBRARY=other.so python tools/dev/equal3d.py /tmp/b.npz; python tools/dev/equal3d.py /tmp/a.npz /tmp/b.npz"""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np
if len(sys.argv) == 3:
    a, b = np.load(sys.argv[1]), np.load(sys.argv[2])
    ok = True
    for k in a.files:
        same = np.array_equal(a[k], b[k]) and np.isfinite(a[k]).all()
        ok &= same
        print("%-16s %s  equal bit for bit: %s" % (k, a[k].shape, same))
    sys.exit(0 if ok else 1)
import bench
from openlbmpm_amd.rk3d import RK3DSlab
out = {}
n = (192, 200, 160)          # (ny no multiple of the tile: a cut tile row)
dom = bench.c5_domain(n)
for state in ("initial", "mixed"):
    for relax in ("MRT", "SRT"):
        rR, rB = bench.c5_state(dom, 0, n[2], state)
        s = RK3DSlab(dom, 0, n[2], dict(relax=relax))
        s.set_density(rR, rB)
        s.step_single(40); s.sync()
        s.phase_field(diagnostics=True)
        for f in ("rhoR", "rhoB", "phi"):
            out["%s_%s_%s" % (state, relax, f)] = s.get(f)
        s.close()
np.savez(sys.argv[1], **out)
print("written", sys.argv[1])
