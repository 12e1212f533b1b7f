"""dev: where a wave of rk3dq_fused spends a march step (cycle counter at six marks; -DLBMPM_DEV -DLBMPM_PHASES build)"""
import sys, os, ctypes as C
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
os.environ["LBMPM_RK3D_TRACE"] = "1"
if not os.environ.get("LBMPM_LIBRARY"):          # build the -DLBMPM_DEV -DLBMPM_PHASES flavour beside the tools (hipcc needed) and load it
    from openlbmpm_amd import build
    out = os.path.join(ROOT, "tools", "dev", "_build", "liblbmpm_hip_phases.so")
    if not os.path.exists(out) or any(os.path.getmtime(d) > os.path.getmtime(out) for d in build.sources()):
        build.build_dev(out, extra=("-DLBMPM_PHASES",))
    os.environ["LBMPM_LIBRARY"] = out
import numpy as np
from openlbmpm_amd.rk3d import RK3DSlab
import bench
n = 512
dom = bench.c5_domain((n, n, n))
for state in ("initial", "mixed"):
    rR, rB = bench.c5_state(dom, 0, n, state)
    s = RK3DSlab(dom, 0, n, dict(relax="MRT"))
    s.set_density(rR, rB)
    s.step_single(4); s.sync()
    nb = 1 << 19
    out = np.zeros(nb, dtype=np.uint64)                    # 4 MB trace area
    s._L.lbmpm_rk3d_debug_trace(s._h, out.ctypes.data_as(C.c_void_p), nb // 4)
    a = out[(1 << 18):(1 << 18) + 1024 * 64].reshape(1024, 8, 8).astype(np.float64)
    ok = a[:, 0, 6] > 0
    a = a[ok]
    per = a[:, :, :6] / a[:, :, 6:7]                       # cycles per march step, per workgroup and wave
    names = ["top + rim", "own cells", "put_s + wait", "issue pulls", "barrier", "collide + stores"]
    print(state, "workgroups", int(ok.sum()), " cycles per march step (s_memtime ticks), median over workgroups, by wave 0..7:")
    for k, nm in enumerate(names):
        print("  %-18s" % nm, " ".join("%6.0f" % v for v in np.median(per[:, :, k], axis=0)))
    print("  %-18s" % "sum", " ".join("%6.0f" % v for v in np.median(per.sum(axis=2), axis=0)))
    s.close()
