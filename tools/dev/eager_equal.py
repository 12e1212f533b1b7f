"""dev: the bench lattices c2, c4 stepped through whatever library LBMPM_LIBRARY names; the final populations (and tracer) are written to
argv[1].  Run once with the product library and once with a -DRK2D_NO_EAGER build (plain C++ pulls), then compare the two files bit for bit:
    python -c "from openlbmpm_amd import build; build.build_dev('tools/dev/_build/noeager.so', extra=('-DRK2D_NO_EAGER',))"
    python tools/dev/eager_equal.py /tmp/a.npz; LBMPM_LIBRARY=tools/dev/_build/noeager.so python tools/dev/eager_equal.py /tmp/b.npz; python tools/dev/eager_equal.py /tmp/a.npz /tmp/b.npz"""
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
for name, build, size, steps in (("c2", bench.build_c2, 1000, 300), ("c4", bench.build_c4, 1500, 300)):
    s, _, _ = build(size, size + 37, 0)          # (sizes that are no multiple of the tile: partial tiles on two edges)
    s.enable_diagnostics(True)
    s.step(steps); s.sync()
    for f in ("rhoR", "rhoB", "vx", "vy"):
        out["%s_%s" % (name, f)] = s.get(f)
    if name == "c4":
        out["c4_C"] = s.get_tracer(0)
    s.close()
np.savez(sys.argv[1], **out)
print("written", sys.argv[1])
