"""Multi-GPU sharding of ONE world: basins over ranks, one gather of the outputs at write-out.

The reference has no distributed path (its only parallelism is a joblib thread pool over basin chunks,
abcd.py:357-391).  What shards naturally on this path:

* Penman-Monteith: every (cell, month) is independent given ``tairprev`` (the previous CELL's temperature, an input);
* ABCD: cells are independent except for the per-basin means after spin-up (abcd.py:274-278) -> whole basins per rank;
* MRTM: independent per river network -> whole networks per rank.

So the unit of sharding is a connected component of "same basin OR linked by a flow edge".  Components are packed onto
the ranks largest-first onto the least-loaded rank (LPT), each rank runs the unchanged single-GPU pipeline on its
cells (no collective on the data path), and the six ``[n_local, nmonths]`` outputs travel to rank 0 in ONE padded
gather over RCCL/xGMI (``torch.distributed`` backend "nccl"; "gloo" in the CPU tests), where rows are scattered back to
grid order.
"""
from types import SimpleNamespace

import numpy as np

from .routing.mrtm import UpstreamMatrix


def shard_components(basin_ids, um):
    """Label cells by connected component of (same basin) U (flow edge). Returns labels [ncell] (0..k-1)."""
    basin_ids = np.asarray(basin_ids)
    n = len(basin_ids)
    parent = np.arange(n)

    def find(x):
        while parent[x] != x:
            parent[x] = parent[parent[x]]
            x = parent[x]
        return x
    first = {}
    for c, b in enumerate(basin_ids):
        if b in first:
            ra, rb = find(c), find(first[b])
            if ra != rb:
                parent[max(ra, rb)] = min(ra, rb)
        else:
            first[b] = c
    rows = np.repeat(np.arange(n), np.diff(um.indptr))
    for r, c in zip(rows, um.indices):
        if r != c:
            ra, rb = find(int(r)), find(int(c))
            if ra != rb:
                parent[max(ra, rb)] = min(ra, rb)
    roots = np.array([find(c) for c in range(n)])
    _, labels = np.unique(roots, return_inverse=True)
    return labels


def make_shards(world, um, n_ranks):
    """LPT packing of the components onto ``n_ranks``. Returns a list of shards (namespace: rank, cells)."""
    labels = shard_components(world.basin_ids, um)
    sizes = np.bincount(labels)
    load = np.zeros(n_ranks, dtype=np.int64)
    owner = np.empty(len(sizes), dtype=np.int64)
    for comp in np.argsort(-sizes, kind='stable'):
        r = int(np.argmin(load))
        owner[comp] = r
        load[r] += sizes[comp]
    cell_rank = owner[labels]
    return [SimpleNamespace(rank=r, cells=np.nonzero(cell_rank == r)[0].astype(np.int64)) for r in range(n_ranks)]


def sub_matrix(um, cells):
    """UM restricted to ``cells`` (closed under flow edges), re-indexed to 0..len(cells)-1."""
    cells = np.asarray(cells, dtype=np.int64)
    new_of = np.full(um.shape[0], -1, dtype=np.int64)
    new_of[cells] = np.arange(len(cells))
    counts = np.diff(um.indptr)[cells]
    indptr = np.concatenate([[0], np.cumsum(counts)])
    take = np.concatenate([np.arange(um.indptr[c], um.indptr[c + 1]) for c in cells]) if len(cells) else np.zeros(0, int)
    cols = new_of[um.indices[take]]
    if (cols < 0).any():
        raise ValueError('shard is not closed under flow edges')
    return UpstreamMatrix(indptr, cols, um.sign[take])


def sub_world(world, um, shard):
    """The shard's slice of the world (same attribute names) and its routing matrix."""
    c = shard.cells
    w = SimpleNamespace(**vars(world))
    w.ncell = len(c)
    for name in ('coords', 'basin_ids', 'flow_dir', 'area', 'flow_dist', 'velocity', 'elev', 'latitude', 'lct'):
        setattr(w, name, getattr(world, name)[c])
    return w, sub_matrix(um, c)


def fill_shard_forcing(ctx, world, shard, pipe, seed, nan_frac=0.0):
    """Generate the whole world's forcing on this device, keep the shard's rows (+ tairprev rows of cell - 1)."""
    nm = pipe.nmonths
    full = {k: ctx.empty((world.ncell, nm)) for k in pipe.alloc_forcing()}
    d_lat = ctx.upload(world.latitude)
    ctx.synth_forcing(seed, world.ncell, nm, d_lat, full, nan_frac=nan_frac)
    d_rows = ctx.upload(shard.cells, dtype=np.int64)
    for k, dst in pipe.forcing.items():
        ctx.gather_rows(full[k], d_rows, len(shard.cells), nm, dst)
    # tairprev[c] = tas[c - 1] in GLOBAL cell order, zeros for cell 0 (data_load.py:128-129)
    prev = np.maximum(shard.cells - 1, 0)
    d_prev = ctx.upload(prev, dtype=np.int64)
    pipe.d_tairprev = ctx.empty((len(shard.cells), nm))
    ctx.gather_rows(full['tas'], d_prev, len(shard.cells), nm, pipe.d_tairprev)
    if len(shard.cells) and shard.cells[0] == 0:
        from . import _hip
        ctx._check(_hip.lib().xh_memset(ctx.handle, pipe.d_tairprev.ptr, 0, nm * 8))
    ctx.sync()
    for b in list(full.values()) + [d_lat, d_rows, d_prev]:
        b.free()


def gather_to_root(local, shards, ncell, dist, root=0):
    """ONE gather of a stacked ``[nvar, n_local, ncols]`` tensor to ``root``; returns ``[nvar, ncell, ncols]`` there.

    ``local`` is a torch tensor on the rank's device (CUDA with the "nccl" backend = RCCL, CPU with "gloo").
    Shards differ in size, so every rank pads to the largest shard (RCCL has no gather-v).
    """
    import torch
    rank, world_size = dist.get_rank(), dist.get_world_size()
    nvar, n_local, ncols = local.shape
    n_max = max(len(s.cells) for s in shards)
    send = local
    if n_local != n_max:
        send = torch.zeros((nvar, n_max, ncols), dtype=local.dtype, device=local.device)
        send[:, :n_local] = local
    recv = [torch.empty_like(send) for _ in range(world_size)] if rank == root else None
    dist.gather(send.contiguous(), gather_list=recv, dst=root)
    if rank != root:
        return None
    out = torch.empty((nvar, ncell, ncols), dtype=local.dtype, device=local.device)
    for s, buf in zip(shards, recv):
        idx = torch.as_tensor(s.cells, device=local.device)
        out.index_copy_(1, idx, buf[:, :len(s.cells)])
    return out


def gather_outputs(ctx, pipe, shard, shards, world, dist, torch, names=('pet', 'aet', 'q', 'sav', 'chs', 'avg')):
    """Copy the pipeline's outputs into one torch CUDA tensor and gather them on rank 0 (grid order)."""
    from . import _hip
    n, nm = pipe.ncell, pipe.nmonths
    local = torch.empty((len(names), n, nm), dtype=torch.float64, device='cuda')
    for i, k in enumerate(names):
        ctx._check(_hip.lib().xh_memcpy_d2d(ctx.handle, local[i].data_ptr(), pipe.out[k].ptr, n * nm * 8))
    ctx.sync()                       # our stream -> torch's stream hand-off
    if dist.get_backend() != 'nccl':
        local = local.cpu()          # gloo dry runs gather on the host
    return gather_to_root(local, shards, world.ncell, dist)
