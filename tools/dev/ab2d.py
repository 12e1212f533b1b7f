"""dev: A/B of library builds on the 2-D bench lattices (one box, alternating):  python tools/dev/ab2d.py [rounds=2] libA.so libB.so ...
every build in its own process (LBMPM_LIBRARY); c2 1024^2, c3 2048^2 porous, c4 2048^2 porous + tracer; ms per step, best of 3 x 300 steps"""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
if len(sys.argv) > 1 and sys.argv[1] == "--child":
    sys.path.insert(0, ROOT)
    import bench
    out = []
    for name, build, size in (("c2", bench.build_c2, (1024, 1024)), ("c3", bench.build_c3, (2048, 2048)), ("c4", bench.build_c4, (2048, 2048))):
        s, _, _ = build(size[0], size[1], 0)
        s.step(100); s.sync()
        best = min(s.step_timed(300)[1] / 300 for _ in range(3))
        out.append("%s %.4f ms" % (name, best))
        s.close()
    print(" | ".join(out), flush=True)
    sys.exit(0)
args = sys.argv[1:]
rounds = int(args.pop(0)) if args and args[0].isdigit() else 2
for r in range(rounds):
    for lib in args:
        p = subprocess.run([sys.executable, os.path.abspath(__file__), "--child"], env=dict(os.environ, LBMPM_LIBRARY=os.path.abspath(lib)), capture_output=True, text=True, timeout=900)
        print("%-24s %s" % (os.path.basename(lib), p.stdout.strip() or p.stderr.strip()[-300:]), flush=True)
