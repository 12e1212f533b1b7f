#!/usr/bin/env python3
"""Quick kernel-schedule comparison on one GPU (tuning aid, not the headline bench)."""
import os, sys, subprocess, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
cases = sys.argv[1:] or ["v1", "s0", "s1", "s2", "s3", "s4"]
for c in cases:
    env = dict(os.environ)
    args = ["--variant", "0"]
    if c.startswith("v"):
        args = ["--variant", c[1:]]
    else:
        env["LBMPM_RK2D_SHAPE"] = c[1:]      # a knob of the development build (openlbmpm_amd/build.py::build_dev_if_stale)
        sys.path.insert(0, ROOT)
        from openlbmpm_amd import build
        env["LBMPM_LIBRARY"] = build.build_dev_if_stale(verbose=False)
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--no-cpu-baseline", "--steps", "1000",
                          "--warmup", "100"] + args, env=env, capture_output=True, text=True)
    try:
        j = json.loads(out.stdout.strip().splitlines()[-1])
        print(c, "MLUPS %.0f  ms/step %.4f  dom_kernel_ms %.4f  frac %.3f" % (
            j["value"], j["ms_per_step"], j["roofline"]["avg_launch_ms"], j["roofline"]["frac"]), flush=True)
    except Exception as e:
        print(c, "FAILED", out.stderr[-400:])
