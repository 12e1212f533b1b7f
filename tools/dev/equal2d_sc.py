"""dev: c3 (and a small SC case) stepped through whatever library LBMPM_LIBRARY names; populations' densities written to argv[1]; two files: compare"""
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
        print("%-12s %s  equal bit for bit: %s" % (k, a[k].shape, same))
    sys.exit(0 if ok else 1)
import bench
out = {}
s, _, _ = bench.build_c3(1500, 1537, 0)
s.enable_diagnostics(True)
s.step(300); s.sync()
for f in ("f0", "f1", "rho0", "rho1", "vx", "vy"):
    out["c3_" + f] = s.get(f)
s.close()
np.savez(sys.argv[1], **out)
print("written", sys.argv[1], list(out))
