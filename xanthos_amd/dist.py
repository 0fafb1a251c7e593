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
the process group -- any object with ``rank``, ``size``, ``bcast``, ``allreduce`` and ``gather`` like
``launch.SocketGroup``, this package's own TCP rendezvous; ``bench.py`` wraps ``torch.distributed`` the same way -- only
carries the RCCL ids and a flag.  ``host_gather`` (the rows through the group itself) is the fall-back when RCCL cannot make
a communicator, e.g. a dry run with every rank on one GPU.  No PyTorch in this package.
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
    if um is not None:                       # (no routing selected: whole basins are all the ranks need)
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


def assemble_rows(parts, shards, ncell):
    """Rank-major blocks ``[nvar, n_local(rank), ncols]`` -> ``[nvar, ncell, ncols]`` in grid order (rows of rank r are the
    cells ``shards[r].cells``)."""
    nvar, ncols = parts[0].shape[0], parts[0].shape[2]
    out = np.empty((nvar, ncell, ncols))
    for part, sh in zip(parts, shards):
        out[:, sh.cells, :] = part
    return out


def host_gather(arrays, shards, ncell, group, root=0):
    """Fall-back of the write-out gather when RCCL is not available (two test ranks on one GPU, CPU dry runs): every rank's
    ``[nvar, n_local, ncols]`` host block travels through the process group (``group.gather(..., raw=True)``: one message
    per rank) and the root reorders the rows.  Returns ``[nvar, ncell, ncols]`` on ``root``, None elsewhere."""
    local = np.ascontiguousarray(arrays, dtype=np.float64)
    got = group.gather(memoryview(local).cast('B'), root=root, raw=True)
    if group.rank != root:
        return None
    nvar, ncols = local.shape[0], local.shape[2]
    parts = [np.frombuffer(b, dtype=np.float64).reshape(nvar, len(sh.cells), ncols) for b, sh in zip(got, shards)]
    return assemble_rows(parts, shards, ncell)


class OutputGather:
    """The write-out gather of a sharded run, set up once and run with every pipeline pass.

    kind "rccl": xh_comm_gather_rows on the library's streams -- no torch tensor, no staging copy on the senders, no
    padding; rank 0 ends up with ``names`` as device arrays ``[ncell, nmonths]`` in grid order (``self.out``).  Two phases:
    ``run_side`` sends PET / AET / Q / Sav on the context's gather stream as soon as they are final (pass it to
    ``DevicePipeline.run(after_runoff=...)``), i.e. beside the routing kernel, which needs none of them;
    ``run_tail`` sends ChStorage / Avg_ChFlow behind the routing and joins the two.  What a step still spends in the gather
    after the routing has ended is timed on the device (``exposed_ms``).  One communicator per stream.
    kind "host": the rows through the process group (``host_gather``; dry runs, RCCL unavailable), all in ``run_tail``; the root
    uploads the assembled arrays, so ``self.out`` holds device arrays in grid order either way."""

    SIDE = ('pet', 'aet', 'q', 'sav')

    def __init__(self, ctx, pipe, shards, group, ncell, names=('pet', 'aet', 'q', 'sav', 'chs', 'avg'), root=0):
        from . import _hip
        rank = group.rank
        self.ctx, self.pipe, self.shards, self.rank, self.ncell = ctx, pipe, shards, rank, int(ncell)
        self.group, self.names, self.root = group, tuple(names), root
        self.side_names = tuple(k for k in self.names if k in self.SIDE)
        self.tail_names = tuple(k for k in self.names if k not in self.SIDE)
        self.counts = np.array([len(s.cells) for s in shards], dtype=np.int64)
        self.out, self.d_perm, self.comm, self.comm_side, self.kind, self.why = None, None, None, None, 'host', ''
        self.side_done, self.runs, self._last = False, 0, None
        self.bytes = int(self.counts.sum()) * pipe.nmonths * 8 * len(self.names)
        # The library's own gather (RCCL bound at run time) whatever backend the launcher's process group uses: the group
        # only carries the 128-byte ids and the agreement below (with "gloo" on host tensors -- the dry run with every rank
        # on one GPU, where a test-only stand-in for librccl.so.1 may be named by XH_RCCL_LIBRARY).
        # Agree on RCCL availability BEFORE ncclCommInitRank: the init is itself a collective, so a rank that cannot even
        # load librccl must not leave the others blocked inside it.  comm_unique_id() loads the library and makes an id
        # (cheap, local); only the root's ids are used.
        uid, ok = [None, None], 1
        try:
            mine = [_hip.comm_unique_id(), _hip.comm_unique_id()]
            if rank == root:
                uid = mine
        except (_hip.HipError, RuntimeError) as exc:
            self.why, ok = str(exc), 0
        if int(group.allreduce(ok, 'min')) == 1:
            uid = group.bcast([bytes(u) for u in uid] if rank == root else None, src=root)
            try:
                self.comm = _hip.Comm(ctx, len(shards), rank, uid[0])
                self.comm_side = _hip.Comm(ctx, len(shards), rank, uid[1])
                self.kind = 'rccl'
            except (_hip.HipError, RuntimeError) as exc:      # e.g. two ranks on one GPU in a dry run
                self.why = str(exc)
            if int(group.allreduce(1 if self.kind == 'rccl' else 0, 'min')) == 0:      # all ranks take the same path
                for c in (self.comm, self.comm_side):
                    if c is not None:
                        c.close()
                self.comm, self.comm_side, self.kind = None, None, 'host'
        # which library answered: the test-only stand-in of tests/fake_rccl marks its ids (the report must say so)
        self.library = None
        if self.kind == 'rccl':
            self.library = 'test stand-in (tests/fake_rccl)' if bytes(uid[0][:4]) == b'FAKE' else 'librccl.so.1'
        if rank == root:
            if self.kind == 'rccl':
                self.d_perm = ctx.upload(np.concatenate([s.cells for s in shards]), dtype=np.int64)
            self.out = {k: ctx.empty((self.ncell, pipe.nmonths)) for k in self.names}

    def _gather(self, comm, names, side):
        comm.gather_rows([self.pipe.out[k] for k in names], self.counts, self.pipe.nmonths, perm=self.d_perm,
                         out=None if self.out is None else [self.out[k] for k in names], root=self.root, side=side)

    def run_side(self):
        """PET / AET / Q / Sav, on the gather stream, beside the routing (a no-op for kind "host")."""
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
        local = np.stack([self.pipe.out[k].download() for k in self.names])      # (the download settles a routing fault first)
        got = host_gather(local, self.shards, self.ncell, self.group, root=self.root)
        if got is not None:
            for i, k in enumerate(self.names):
                self.out[k].upload(got[i])
        self.runs += 1
        return None

    def run(self):
        """Everything behind the routing (no overlap): what round 3 did, kept for callers without an after_runoff hook."""
        return self.run_tail()

    def last(self):
        """On the root: the arrays of the last gather as host arrays, by name (grid order)."""
        self.ctx.sync()
        return {k: self.out[k].download() for k in self.names}

    def report(self):
        exposed = None
        if self.kind == 'rccl':
            ms, n = self.ctx.timing('gather_exposed')
            exposed = ms / n if n else None
        return {'kind': self.kind, 'library': self.library, 'bytes_per_step': self.bytes, 'variables': list(self.names),
                'beside_the_routing': list(self.side_names) if self.kind == 'rccl' else [],
                'behind_the_routing': list(self.tail_names) if self.kind == 'rccl' else list(self.names),
                'process_group': type(self.group).__name__,
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
