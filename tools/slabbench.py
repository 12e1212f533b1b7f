#!/usr/bin/env python3
"""Decomposition overhead of the slab path on ONE GPU: k virtual ranks (same kernels, same
interior/boundary schedule, halos moved by device copies) against the single slab.  What is left
for a real k-GPU run on top of this is the xGMI transfer time not hidden under the interior planes."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
import bench
from openlbmpm_amd.rk3d import RK3DCluster

n = int(sys.argv[1]) if len(sys.argv) > 1 else 512
ks = [int(v) for v in sys.argv[2:]] or [1, 2, 4, 8]
dom = bench.c5_domain((n, n, n))
rR, rB = bench.c5_densities(dom, 0, n)
nf = int(dom.sum())
for k in ks:
    c = RK3DCluster(dom, k, dict(relax=os.environ.get("LBMPM_K3_RELAX", "MRT")))
    c.set_density(rR, rB)
    c.step(3)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    steps = 20
    c.step(steps)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / steps
    print("k=%d  %.3f ms per step of the whole lattice  (%.0f MLUPS on one GPU; x%d GPUs would need %.3f ms + exposed transfers)"
          % (k, dt * 1e3, nf / dt / 1e6, k, dt * 1e3 / k), flush=True)
    c.close()
