"""dev: where a wave of rk2d_fused spends a tile (cycle counter at the phase borders, summed over all waves; -DLBMPM_DEV -DLBMPM_PHASES2D build)"""
import sys, os, ctypes as C
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
if not os.environ.get("LBMPM_LIBRARY"):
    from openlbmpm_amd import build
    out = os.path.join(ROOT, "tools", "dev", "_build", "liblbmpm_hip_phases2d.so")
    if not os.path.exists(out) or any(os.path.getmtime(d) > os.path.getmtime(out) for d in build.sources()):
        build.build_dev(out, extra=("-DLBMPM_PHASES2D",))
    os.environ["LBMPM_LIBRARY"] = out
import numpy as np
import bench
from openlbmpm_amd import _lib
names = ["flags + barrier", "pulls -> phi (A)", "barrier", "phi on solids (B)", "gradient, normals (C)", "barrier", "force, collision, store issue (D)", "stores acknowledged"]
for name, build, size in (("c2", bench.build_c2, 1024), ("c2@2048", bench.build_c2, 2048), ("c4", bench.build_c4, 2048)):
    s, _, _ = build(size, size, 0)
    s.step(20); s.sync()
    buf = (C.c_ulonglong * 16)()
    L = _lib.lib()
    L.lbmpm_dev_rk2d_phases(buf)
    s.step(50); s.sync()
    L.lbmpm_dev_rk2d_phases(buf)
    a = np.array(list(buf), dtype=np.float64)
    n = a[15]
    print("%s: %d waves per step; cycles per wave and tile (s_memtime ticks):" % (name, n / 50))
    for k, nm in enumerate(names):
        print("   %-36s %8.0f" % (nm, a[k] / n))
    print("   %-36s %8.0f" % ("sum", a[:8].sum() / n))
    print("   %-36s %8.0f" % ("(of D: the tracer sub-step)", a[8] / n))
    s.close()
