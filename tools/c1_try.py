import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
for env in ("", "1"):
    if env: os.environ["LBMPM_NO_GRAPH"] = "1"
    s, _, _ = bench.build_c1(128, 128, 0)
    w, mt, md = bench.time_solver_2d(s, 2000, 200)
    print("NO_GRAPH=%r  %.3f us/step wall, %.3f us/step device, %.0f MLUPS" % (env, w * 1e6 / 2000, mt * 1e3 / 2000, 16384 * 2000 / w / 1e6), flush=True)
    s.close()
