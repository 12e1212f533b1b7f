import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import bench
s, _, _ = bench.build_c1(128, 128, 0)
w, mt, md = bench.time_solver_2d(s, 20000, 2000)
print("c1 %s: %.4f us per step (wall %.4f), %.0f MLUPS" % ("one launch per step (hipGraph of 64)", md / 20000 * 1e3, w / 20000 * 1e6, s.num_fluid_nodes * 20000 / w / 1e6))
s.close()
