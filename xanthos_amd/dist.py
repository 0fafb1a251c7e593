"""Multi-GPU sharding of ONE world: basins over ranks, one gather of the outputs at write-out.

The reference has no distributed path (its only parallelism is a joblib thread pool over basin chunks,
abcd.py:357-391).  What shards naturally on this path:

* Penman-Monteith: every (cell, month) is independent given ``tairprev`` (the previous CELL's temperature, an input);
* ABCD: cells are independent except for the per-basin means after spin-up (abcd.py:274-278) -> whole basins per rank;
* MRTM: independent per river network -> whole networks per rank.

So the unit of sharding is a connected component of "same basin OR linked by a flow edge".  Components are packed onto
the ranks largest-first onto the least-loaded rank (LPT), each rank runs the unchanged single-GPU pipeline on its
cells (no collective on the data path), and the six ``[n_local, nmonths]`` outputs travel to rank 0 in ONE gather over
RCCL/xGMI, where rows are scattered back to grid order.  The gather is the library's own (``xh_comm_gather_rows``:
grouped ncclSend / ncclRecv of the exact shard sizes straight from the pipeline's output buffers, csrc/xh_comm.hip);
``torch.distributed`` is only the launcher and carries the RCCL id.  ``gather_to_root`` (a padded
``torch.distributed.gather``) remains for the gloo CPU tests and as the fallback when RCCL cannot be initialised
(e.g. the dry run with every rank on one GPU).
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
    """Generate THIS shard's rows of the world's forcing on the device (the random streams are keyed on the global
    cell index, so the rows equal those of a whole-world generation) and its tairprev rows = temperature of the
    previous GLOBAL cell, zeros for cell 0 (data_load.py:128-129)."""
    nm, n = pipe.nmonths, len(shard.cells)
    forcing = pipe.alloc_forcing()
    d_lat = ctx.upload(world.latitude[shard.cells])
    d_ids = ctx.upload(shard.cells, dtype=np.int64)
    ctx.synth_forcing(seed, n, nm, d_lat, forcing, nan_frac=nan_frac, cell_ids=d_ids)
    prev = shard.cells - 1                                    # -1 for global cell 0: a row of zeros
    d_prev = ctx.upload(prev, dtype=np.int64)
    d_plat = ctx.upload(world.latitude[np.maximum(prev, 0)])
    pipe.d_tairprev = ctx.empty((n, nm))
    ctx.synth_forcing(seed, n, nm, d_plat, {'tas': pipe.d_tairprev}, nan_frac=nan_frac, cell_ids=d_prev)
    ctx.sync()
    for b in (d_lat, d_ids, d_prev, d_plat):
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


class OutputGather:
    """The write-out gather of a sharded run, set up once and run with every pipeline pass.

    kind "rccl": xh_comm_gather_rows on the library's streams -- no torch tensor, no staging copy on the senders, no
    padding; rank 0 ends up with ``names`` as device arrays ``[ncell, nmonths]`` in grid order (``self.out``).  Two phases:
    ``run_side`` sends PET / AET / Q / Sav on the context's gather stream as soon as they are final (pass it to
    ``DevicePipeline.run(after_runoff=...)``), i.e. beside the routing kernel, which needs none of them;
    ``run_tail`` sends ChStorage / Avg_ChFlow behind the routing and joins the two.  What a step still spends in the gather
    after the routing has ended is timed on the device (``exposed_ms``).  One communicator per stream.
    kind "torch": padded ``torch.distributed.gather`` of a stacked copy (gloo dry runs; RCCL unavailable), all in ``run_tail``."""

    SIDE = ('pet', 'aet', 'q', 'sav')

    def __init__(self, ctx, pipe, shards, rank, ncell, dist, torch, names=('pet', 'aet', 'q', 'sav', 'chs', 'avg'),
                 root=0):
        from . import _hip
        self.ctx, self.pipe, self.shards, self.rank, self.ncell = ctx, pipe, shards, rank, int(ncell)
        self.dist, self.torch, self.names, self.root = dist, torch, tuple(names), root
        self.side_names = tuple(k for k in self.names if k in self.SIDE)
        self.tail_names = tuple(k for k in self.names if k not in self.SIDE)
        self.counts = np.array([len(s.cells) for s in shards], dtype=np.int64)
        self.out, self.d_perm, self.comm, self.comm_side, self.kind, self.why = None, None, None, None, 'torch', ''
        self.side_done, self.runs, self._last = False, 0, None
        self.bytes = int(self.counts.sum()) * pipe.nmonths * 8 * len(self.names)
        # The library's own gather (RCCL bound at run time) whatever backend the launcher's process group uses: the group
        # only carries the 128-byte ids and the agreement below (with "gloo" on host tensors -- the dry run with every rank
        # on one GPU, where a test-only stand-in for librccl.so.1 may be named by XH_RCCL_LIBRARY).
        # Agree on RCCL availability BEFORE ncclCommInitRank: the init is itself a collective, so a rank that cannot even
        # load librccl must not leave the others blocked inside it.  comm_unique_id() loads the library and makes an id
        # (cheap, local); only the root's ids are used.
        dev = 'cuda' if dist.get_backend() == 'nccl' else 'cpu'
        uid, ok = [None, None], 1
        try:
            mine = [_hip.comm_unique_id(), _hip.comm_unique_id()]
            if rank == root:
                uid = mine
        except (_hip.HipError, RuntimeError) as exc:
            self.why, ok = str(exc), 0
        flag = torch.tensor([ok], device=dev)
        dist.all_reduce(flag, op=dist.ReduceOp.MIN)
        if int(flag.item()) == 1:
            dist.broadcast_object_list(uid, src=root)
            try:
                self.comm = _hip.Comm(ctx, len(shards), rank, uid[0])
                self.comm_side = _hip.Comm(ctx, len(shards), rank, uid[1])
                self.kind = 'rccl'
            except (_hip.HipError, RuntimeError) as exc:      # e.g. two ranks on one GPU in a dry run
                self.why = str(exc)
            flag = torch.tensor([1 if self.kind == 'rccl' else 0], device=dev)
            dist.all_reduce(flag, op=dist.ReduceOp.MIN)       # all ranks take the same path
            if int(flag.item()) == 0:
                for c in (self.comm, self.comm_side):
                    if c is not None:
                        c.close()
                self.comm, self.comm_side, self.kind = None, None, 'torch'
        # which library answered: the test-only stand-in of tests/fake_rccl marks its ids (the report must say so)
        self.library = None
        if self.kind == 'rccl':
            self.library = 'test stand-in (tests/fake_rccl)' if bytes(uid[0][:4]) == b'FAKE' else 'librccl.so.1'
        if self.kind == 'rccl' and rank == root:
            self.d_perm = ctx.upload(np.concatenate([s.cells for s in shards]), dtype=np.int64)
            self.out = {k: ctx.empty((self.ncell, pipe.nmonths)) for k in self.names}

    def _gather(self, comm, names, side):
        comm.gather_rows([self.pipe.out[k] for k in names], self.counts, self.pipe.nmonths, perm=self.d_perm,
                         out=None if self.out is None else [self.out[k] for k in names], root=self.root, side=side)

    def run_side(self):
        """PET / AET / Q / Sav, on the gather stream, beside the routing (a no-op for kind "torch")."""
        if self.kind == 'rccl' and self.side_names:
            self._gather(self.comm_side, self.side_names, True)
            self.side_done = True

    def run_tail(self):
        """ChStorage / Avg_ChFlow behind the routing (and whatever run_side did not send), then the join."""
        if self.kind == 'rccl':
            self.ctx.sync()       # settles a routing fault (re-route) before ChStorage / Avg_ChFlow leave (it also waits for
                                  # the side gather: the two share the links anyway)
            self.ctx.mark_begin('gather_exposed')
            rest = self.tail_names if self.side_done else self.names
            if rest:
                self._gather(self.comm, rest, False)
            self.ctx.comm_join()
            self.ctx.mark_end()
            self.side_done = False
            self.runs += 1
            return None
        world = SimpleNamespace(ncell=self.ncell)
        got = gather_outputs(self.ctx, self.pipe, self.shards[self.rank], self.shards, world, self.dist, self.torch,
                             names=self.names)
        if self.torch.cuda.is_available() and self.dist.get_backend() == 'nccl':
            self.torch.cuda.synchronize()
        self._last = got
        self.runs += 1
        return got

    def run(self):
        """Everything behind the routing (no overlap): what round 3 did, kept for callers without an after_runoff hook."""
        return self.run_tail()

    def last(self):
        """On the root: the arrays of the last gather as host arrays, by name (grid order)."""
        if self.kind == 'rccl':
            self.ctx.sync()
            return {k: self.out[k].download() for k in self.names}
        return {k: self._last[i].cpu().numpy() for i, k in enumerate(self.names)}

    def report(self):
        exposed = None
        if self.kind == 'rccl':
            ms, n = self.ctx.timing('gather_exposed')
            exposed = ms / n if n else None
        return {'kind': self.kind, 'library': self.library, 'bytes_per_step': self.bytes, 'variables': list(self.names),
                'beside_the_routing': list(self.side_names) if self.kind == 'rccl' else [],
                'behind_the_routing': list(self.tail_names) if self.kind == 'rccl' else list(self.names),
                'exposed_ms': exposed, 'exposed_ms_is': 'device time of a step between the end of the routing kernel and '
                'the end of the gather (HIP events on the context stream), mean over the timed steps',
                'rows_per_rank': self.counts.tolist(), 'fallback_reason': self.why}

    def close(self):
        for c in (self.comm, self.comm_side):
            if c is not None:
                self.ctx.sync()
                c.close()
        self.comm = self.comm_side = None
        for a in (list(self.out.values()) if self.out else []) + ([self.d_perm] if self.d_perm is not None else []):
            a.free()
        self.out, self.d_perm = None, None
