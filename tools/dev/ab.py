"""dev: A/B of library builds on the c5 bench lattice (one box, back to back, alternating):
    python tools/dev/ab.py [n=512] [rounds=2] libA.so libB.so ...
every build in its own process (LBMPM_LIBRARY), the bench's initial state and the state with both colours in every cell, MRT and SRT"""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
if len(sys.argv) > 1 and sys.argv[1] == "--child":
    sys.path.insert(0, ROOT)
    import numpy as np
    import bench
    from openlbmpm_amd.rk3d import RK3DSlab
    n, relaxes = int(sys.argv[2]), sys.argv[3].split(",")
    dom = bench.c5_domain((n, n, n))
    out = []
    for state in ("initial", "mixed"):
        rR, rB = bench.c5_state(dom, 0, n, state)
        for relax in relaxes:
            s = RK3DSlab(dom, 0, n, dict(relax=relax))
            s.set_density(rR, rB)
            s.step_single(4); s.sync()
            best = min(s.step_timed(10)[0] / 10 for _ in range(3))
            out.append("%s/%s %.3f ms %.0f MLUPS" % (state, relax, best, s.num_fluid_nodes / best / 1e3))
            s.close()
    print(" | ".join(out), flush=True)
    sys.exit(0)
args = sys.argv[1:]
n = int(args.pop(0)) if args and args[0].isdigit() else 512
rounds = int(args.pop(0)) if args and args[0].isdigit() else 2
relaxes = os.environ.get("AB_RELAX", "MRT")
for r in range(rounds):
    for lib in args:
        env = dict(os.environ, LBMPM_LIBRARY=os.path.abspath(lib))
        p = subprocess.run([sys.executable, os.path.abspath(__file__), "--child", str(n), relaxes], env=env, capture_output=True, text=True, timeout=900)
        print("%-28s %s" % (os.path.basename(lib), p.stdout.strip() or p.stderr.strip()[-300:]), flush=True)
