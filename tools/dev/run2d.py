"""dev: a few steps of c3 and c4 (for counter passes):  python tools/dev/run2d.py [steps=20] [workloads=c3,c4]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import bench
steps = int(sys.argv[1]) if len(sys.argv) > 1 else 20
wls = (sys.argv[2] if len(sys.argv) > 2 else "c3,c4").split(",")
B = {"c2": (bench.build_c2, 1024), "c3": (bench.build_c3, 2048), "c4": (bench.build_c4, 2048)}
for w in wls:
    s, _, _ = B[w][0](B[w][1], B[w][1], 0)
    s.step(steps); s.sync()
    print(w, "fluid", s.num_fluid_nodes, flush=True)
    s.close()
