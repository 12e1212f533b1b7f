#!/usr/bin/env python3
"""Is a face message complete?  pack N times into zeroed buffers; the set of zero entries must not change."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np, torch
import bench
from openlbmpm_amd.rk3d import RK3DCluster
n = int(sys.argv[1]) if len(sys.argv) > 1 else 192
K = int(sys.argv[2]) if len(sys.argv) > 2 else 3
dom = bench.c5_domain((n, n, n)); rR, rB = bench.c5_densities(dom, 0, n)
c = RK3DCluster(dom, K, dict(relax="MRT")); c.set_density(rR, rB)
for name in ("f_send_up", "f_send_down"):
    for r, s in enumerate(c.slabs):
        t = s.buffer(name); ref = None
        for it in range(6):
            with torch.cuda.stream(c.stream):
                t.zero_(); s.pack()
            c.stream.synchronize()
            z = (t == 0).cpu().numpy()
            if ref is None: ref = z; print("rank %d %s: %d doubles, %d zero" % (r, name, z.size, int(z.sum())))
            elif not np.array_equal(z, ref):
                d = np.flatnonzero(z != ref); print("   pack %d: %d entries differ: %s ..." % (it, d.size, d[:8].tolist()))
# the same through the copy and the unpack
with torch.cuda.stream(c.stream):
    for s in c.slabs: s.pack()
c.stream.synchronize()
for r in range(K - 1):
    a, b = c.slabs[r].buffer("f_send_up"), c.slabs[r + 1].buffer("f_recv_below")
    print("sizes up %d->%d: %d %d" % (r, r + 1, a.numel(), b.numel()))
    a, b = c.slabs[r + 1].buffer("f_send_down"), c.slabs[r].buffer("f_recv_above")
    print("sizes down %d->%d: %d %d" % (r + 1, r, a.numel(), b.numel()))
for it in range(6):
    with torch.cuda.stream(c.stream):
        for s in c.slabs:
            for nm in ("f_recv_below", "f_recv_above"): s.buffer(nm).zero_()
        c._exchange("f")
    c.stream.synchronize()
    for r in range(K - 1):
        for a, b in ((c.slabs[r].buffer("f_send_up"), c.slabs[r + 1].buffer("f_recv_below")), (c.slabs[r + 1].buffer("f_send_down"), c.slabs[r].buffer("f_recv_above"))):
            if not torch.equal(a, b): print("  exchange %d: copy differs in %d entries" % (it, int((a != b).sum())))
print("done")
