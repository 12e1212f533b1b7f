#!/usr/bin/env python3
"""K OS processes sharing ONE GPU step the bench lattice through RK3DDistributed -- the in-library IPC transport by default (gloo only
carries the handles at set-up; LBMPM_TRANSPORT=callback: every message staged through the host, RCCL refuses two ranks on one device) --
and rank 0 compares the gathered phase field with the single domain's, bit for bit.

    python -m torch.distributed.run --nnodes=1 --nproc-per-node K --master-addr 127.0.0.1 --master-port P tools/dev/slab_procs.py [n=256] [steps=6]
"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np
import torch, torch.distributed as dist
import bench
from openlbmpm_amd.rk3d import RK3DDistributed, RK3DSlab

n = int(sys.argv[1]) if len(sys.argv) > 1 else 256
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 6
rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
torch.cuda.set_device(0)
dist.init_process_group("gloo")
par = dict(relax=os.environ.get("LBMPM_K3_RELAX", "MRT"))
dom = bench.c5_domain((n, n, n))
rR, rB = bench.c5_densities(dom, 0, n)
d = RK3DDistributed(dom, device=0, params=par)
d.set_density(rR, rB)
warm = int(os.environ.get("SLAB_WARM", "3"))
if warm: d.step(warm)
d.step(steps)
d.observe()
transport = d.timing()["transport"]
phi = d.slab.get("phi")
out = os.environ.get("SLAB_OUT", "/tmp")
np.save(os.path.join(out, "slabprocs_phi_%d.npy" % rank), phi)
d.close()
dist.barrier()
if rank == 0:
    got = np.concatenate([np.load(os.path.join(out, "slabprocs_phi_%d.npy" % r)) for r in range(world)], axis=0)
    ref = RK3DSlab(dom, 0, n, par); ref.set_density(rR, rB); ref.step_single(steps + warm); ref.phase_field(diagnostics=True)
    rphi = ref.get("phi"); ref.close()
    same = bool(np.array_equal(got, rphi))
    print("%d processes, %d^3, %d steps, transport %s: phase field equals the single domain's bit for bit: %s" % (world, n, steps + warm, transport, same))
    if not same:
        bad = ~np.isfinite(got)
        per = bad.reshape(n, -1).sum(axis=1)
        print("  non-finite values per plane:", [(int(z), int(per[z])) for z in np.flatnonzero(per)[:200]])
        for z in np.flatnonzero(per)[:3]:
            ys, xs = np.nonzero(bad[z])
            print("  plane %d: y %d..%d x %d..%d; rows (y: count): %s" % (z, ys.min(), ys.max(), xs.min(), xs.max(), [(int(y), int((ys == y).sum()), int(xs[ys == y].min()), int(xs[ys == y].max())) for y in np.unique(ys)][:80]))
        dd = np.abs(np.nan_to_num(got) - np.nan_to_num(rphi)).reshape(n, -1).max(axis=1)
        print("  planes that differ:", np.flatnonzero(dd > 0)[:100].tolist())
dist.barrier()
dist.destroy_process_group()
