#!/usr/bin/env python3
"""Where does a slab run first differ from the single domain?  Every stored component of every plane (halo planes included),
after the priming exchange and after each step.   python tools/dev/cluster_planes.py [n=192] [k=2] [steps=1]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np
import bench
from openlbmpm_amd.rk3d import RK3DCluster, RK3DSlab
n = int(sys.argv[1]) if len(sys.argv) > 1 else 192
K = int(sys.argv[2]) if len(sys.argv) > 2 else 2
steps = int(sys.argv[3]) if len(sys.argv) > 3 else 1
par = dict(relax=os.environ.get("LBMPM_K3_RELAX", "MRT"))
dom = bench.c5_domain((n, n, n))
rR, rB = bench.c5_densities(dom, 0, n)
c = RK3DCluster(dom, K, par); c.set_density(rR, rB)
ref = RK3DSlab(dom, 0, n, par); ref.set_density(rR, rB)
NAMES = ["g%d" % i for i in range(19)] + ["kR", "Ax", "Ay", "Az", "phi", "flags"]


def compare(tag, planes_of):
    bad = 0
    for r, s in enumerate(c.slabs):
        z0 = c.parts[r][0]
        for zl in planes_of(r, s):
            zg = z0 + zl - 1                      # global plane; the single domain's local index is zg + 1
            if zg < 0 or zg >= n: continue
            fs, fr = s.debug_plane(24, zl), ref.debug_plane(24, zg + 1)
            if not np.array_equal(fs, fr):
                ys, ss = np.nonzero(fs != fr)
                print("  %s rank %d plane %d (global %d): %d row flags differ, e.g. y %d seg %d: %g vs %g" % (tag, r, zl, zg, len(ys), ys[0], ss[0], fs[ys[0], ss[0]], fr[ys[0], ss[0]])); bad += 1
            free = np.repeat((fr == 0), 64, axis=1)[:, :n]         # cells whose records are kept
            comps = range(23) if 1 <= zl <= s.nzl else ([5, 11, 14, 15, 18] if zl == 0 else [6, 12, 13, 16, 17]) + [19, 20, 21, 22]
            for comp in comps:
                a, b = s.debug_plane(comp, zl), ref.debug_plane(comp, zg + 1)
                if comp >= 19: a, b = np.where(free, a, 0.), np.where(free, b, 0.)
                if not np.array_equal(a, b, equal_nan=True):
                    ys, xs = np.nonzero(~((a == b) | (np.isnan(a) & np.isnan(b))))
                    print("  %s rank %d plane %d (global %d) %s: %d cells differ (nan %d), y %d..%d x %d..%d, e.g. (%d,%d): %r vs %r" %
                          (tag, r, zl, zg, NAMES[comp], len(ys), int(np.isnan(a).sum()), ys.min(), ys.max(), xs.min(), xs.max(), ys[0], xs[0], a[ys[0], xs[0]], b[ys[0], xs[0]])); bad += 1
                    if bad > 12: return bad
    print("%s: %s" % (tag, "all equal" if not bad else "%d differences" % bad))
    return bad


halo = lambda r, s: ([0] if r > 0 else []) + ([s.nzl + 1] if r + 1 < K else [])
near = lambda r, s: sorted(set((list(range(0, 12)) if r > 0 else []) + (list(range(s.nzl - 10, s.nzl + 2)) if r + 1 < K else [])))
import torch
with torch.cuda.stream(c.stream):
    c._halo_f()
c.stream.synchronize()
compare("after priming, halo planes", halo)
# halo phase field against the single domain's (diagnostics of the initial state)
ref.phase_field(diagnostics=True); rphi = ref.get("phi")
for r, s in enumerate(c.slabs):
    for zl in halo(r, s):
        zg = c.parts[r][0] + zl - 1
        a = np.where(dom[zg] == 1, s.debug_plane(23, zl), 0.)
        if not np.array_equal(a, rphi[zg]):
            ys, xs = np.nonzero(a != rphi[zg])
            print("  halo phi rank %d plane %d (global %d): %d cells differ, y %d..%d x %d..%d" % (r, zl, zg, len(ys), ys.min(), ys.max(), xs.min(), xs.max()))
        else: print("  halo phi rank %d plane %d: equal" % (r, zl))
for k in range(steps):
    c.step(1); c.stream.synchronize(); ref.step_single(1)
    if compare("after step %d, owned planes near the cuts" % (k + 1), lambda r, s: [z for z in near(r, s) if 1 <= z <= s.nzl]): break
    with torch.cuda.stream(c.stream):
        c._halo_f()
    c.stream.synchronize()
    if compare("after step %d, halo planes" % (k + 1), halo): break
c.close(); ref.close()
