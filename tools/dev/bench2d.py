"""dev: the 2-D bench workloads through whatever library LBMPM_LIBRARY names"""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import bench
tag = os.path.basename(os.environ.get("LBMPM_LIBRARY", "product"))
for name, build, size in (("c2p", bench.build_c2p, 1024), ("c2", bench.build_c2, 1024), ("c2@2048", bench.build_c2, 2048), ("c3", bench.build_c3, 2048), ("c4", bench.build_c4, 2048)):
    s, _, _ = build(size, size, 0)
    w, mt, md = bench.time_solver_2d(s, 600, 60)
    nf = s.num_fluid_nodes
    print("%-10s %-8s %.4f ms/step  %.0f MLUPS  frac %.3f" % (tag, name, md / 600, nf * 600 / w / 1e6, bench.B_ALG[name[:2]] * nf / (md / 600 * 1e-3) / 1e9 / 8000), flush=True)
    s.close()
