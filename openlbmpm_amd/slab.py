"""z-slab decomposition and neighbour halo exchange (host-side plumbing of the 3-D solver).

Pure torch / Python: no lattice arithmetic here.  The same code path serves
  * one process per GPU over torch.distributed (backend "nccl" = RCCL over xGMI) and
  * CPU tensors over gloo in the tests (tests/test_slab_cpu.py).
torch is imported where it is used: a single-slab run never loads it.
"""


def partition_z(nz_global, world):
    """Contiguous z-ranges, remainder planes to the lowest ranks.  The outermost ranks must
    own at least the boundary plane pair (2 planes)."""
    if world < 1 or nz_global < 2 * world:
        raise ValueError("cannot cut %d planes into %d slabs of >= 2 planes" % (nz_global, world))
    base, rem = divmod(nz_global, world)
    out, z = [], 0
    for r in range(world):
        n = base + (1 if r < rem else 0)
        out.append((z, n))
        z += n
    return out


def partition_z_balanced(fluid_per_plane, world, min_planes=2):
    """Contiguous z-ranges with (nearly) equal numbers of FLUID cells: the time of a slab is
    proportional to its fluid cells, and the all-fluid buffer planes at both ends of a porous
    sample would otherwise make the end ranks ~8 % slower than the rest.
    Every rank derives the same cuts from the same mask; no communication."""
    w = [int(v) for v in fluid_per_plane]
    nz = len(w)
    if world < 1 or nz < min_planes * world:
        raise ValueError("cannot cut %d planes into %d slabs of >= %d planes" % (nz, world, min_planes))
    total, cuts, acc, z = float(sum(w)), [0], 0.0, 0
    for r in range(1, world):
        target = total * r / world
        lo, hi = cuts[-1] + min_planes, nz - min_planes * (world - r)
        while z < hi and (z < lo or acc + w[z] * 0.5 <= target):
            acc += w[z]
            z += 1
        cuts.append(z)
    cuts.append(nz)
    parts = [(cuts[r], cuts[r + 1] - cuts[r]) for r in range(world)]
    # a slab without a fluid cell cannot be created (lbmpm_rk3d_create refuses it), and ONE rank raising while the
    # others enter the first halo exchange would hang the job: fail here, on every rank alike (same mask, same cuts)
    empty = [r for r, (z0, n) in enumerate(parts) if sum(w[z0:z0 + n]) == 0]
    if empty:
        even = partition_z(nz, world)
        if all(sum(w[z0:z0 + n]) > 0 for z0, n in even):
            return even
        raise ValueError("slabs %s of %d would hold no fluid cell; use fewer ranks" % (empty, world))
    return parts


def neighbour_exchange(send_up, send_down, recv_from_below, recv_from_above, rank, world, group=None):
    """Every rank sends `send_up` to rank+1 (which receives it in `recv_from_below`) and
    `send_down` to rank-1 (received in `recv_from_above`).  No wrap-around: the global lattice is
    not periodic in z.  Point-to-point only (ncclSend/ncclRecv pairs grouped in one batch), no
    collective on the data path."""
    import torch
    import torch.distributed as dist
    if send_up.is_cuda and dist.get_backend(group) == "gloo":
        # rehearsal transport (several ranks sharing ONE GPU, where RCCL refuses duplicate
        # devices): stage through host memory; same neighbours, same buffers, same order
        torch.cuda.current_stream(send_up.device).synchronize()
        h = [t.cpu() for t in (send_up, send_down, recv_from_below, recv_from_above)]
        neighbour_exchange(h[0], h[1], h[2], h[3], rank, world, group)
        if rank > 0:
            recv_from_below.copy_(h[2])
        if rank + 1 < world:
            recv_from_above.copy_(h[3])
        return
    ops = []
    if rank + 1 < world:
        ops.append(dist.P2POp(dist.isend, send_up, rank + 1, group))
        ops.append(dist.P2POp(dist.irecv, recv_from_above, rank + 1, group))
    if rank > 0:
        ops.append(dist.P2POp(dist.isend, send_down, rank - 1, group))
        ops.append(dist.P2POp(dist.irecv, recv_from_below, rank - 1, group))
    if not ops:
        return
    for req in dist.batch_isend_irecv(ops):
        req.wait()


def local_exchange(slabs_send_up, slabs_send_down, slabs_recv_below, slabs_recv_above):
    """Same data movement between k 'virtual ranks' living in one process (lists indexed by
    virtual rank): used to prove slab-decomposed == single-domain on one GPU."""
    k = len(slabs_send_up)
    for r in range(k):
        if r + 1 < k:
            slabs_recv_below[r + 1].copy_(slabs_send_up[r])
            slabs_recv_above[r].copy_(slabs_send_down[r + 1])


def gather_planes(local, parts, rank, world, group=None, device=None):
    """Stack the ranks' planes on rank 0 (the reference's record is ONE dense array per field, RKD2Q9.py:938-957): `local` is this
    rank's numpy array [n_r, ...], parts = [(z0, n)] of all ranks.  Returns the [nz, ...] array on rank 0, None elsewhere.
    Point-to-point (send / recv to rank 0), through device memory under NCCL, host memory under gloo."""
    import numpy as np
    if world == 1:
        return np.asarray(local)
    import torch
    import torch.distributed as dist
    on_gpu = dist.get_backend(group) == "nccl"
    dev = torch.device("cuda", device if device is not None else torch.cuda.current_device()) if on_gpu else torch.device("cpu")
    peer = (lambda r: dist.get_global_rank(group, r)) if group is not None else (lambda r: r)
    local = np.ascontiguousarray(local)
    if rank != 0:
        dist.send(torch.from_numpy(local).to(dev), dst=peer(0), group=group)
        return None
    tail = local.shape[1:]
    out = np.empty((sum(n for _, n in parts),) + tail, dtype=local.dtype)
    out[parts[0][0]:parts[0][0] + parts[0][1]] = local
    for r in range(1, world):
        z0, n = parts[r]
        buf = torch.empty((n,) + tail, dtype=torch.from_numpy(local[:0]).dtype, device=dev)
        dist.recv(buf, src=peer(r), group=group)
        out[z0:z0 + n] = buf.cpu().numpy()
    return out


class DeviceBuffer:
    """Zero-copy torch view of a raw device allocation owned by liblbmpm_hip.so."""

    def __init__(self, ptr, nbytes):
        self.__cuda_array_interface__ = {"shape": (int(nbytes) // 8,), "typestr": "<f8",
                                         "data": (int(ptr), False), "version": 2}

    def tensor(self, device):
        import torch
        return torch.as_tensor(self, device=device)
