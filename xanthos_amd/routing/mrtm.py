"""Modified River Transport Model -- drop-in for xanthos/routing/mrtm.py on MI355X.

The four reference names are kept (components.py:268-289):

    downstream(coords, flow_dir, settings)          -> dsid [ncell]            (mrtm.py:85-120)
    upstream(coords, dsid, settings)                -> upid [ncell, 9]         (mrtm.py:123-191)
    upstream_genmatrix(upid)                        -> UM                      (mrtm.py:194-230)
    streamrouting(L, S0, F0, ChV, q, area, nday, dt, UM) -> (S, Favg, F)       (mrtm.py:16-82)

plus the whole-series entry the harness prefers, because a call per month would put a launch and two PCIe copies
between months that the persistent kernel keeps on chip:

    route_series(UM, flow_dist, velocity, area, runoff, ndays, spinup_months, S0=None, dt=10800)
        -> (ChStorage, Avg_ChFlow, F_end)                                      (components.py:273-294)

``UM`` is a :class:`UpstreamMatrix`: it carries the CSR triplet of UP - I and lazily builds the device routing plan;
``.tocsr()`` gives the scipy matrix the reference would have produced.  ``streamrouting`` also accepts a scipy
sparse matrix.  The topology functions are host C++ in the same library (csrc/xh_topo.hip).
"""
import numpy as np

from .. import _hip


class UpstreamMatrix:
    """UM = UP - I (mrtm.py:194-230) as CSR arrays + a cached device routing plan."""

    def __init__(self, indptr, indices, sign):
        self.indptr = np.ascontiguousarray(indptr, dtype=np.int64)
        self.indices = np.ascontiguousarray(indices, dtype=np.int32)
        self.sign = np.ascontiguousarray(sign, dtype=np.int8)
        self.shape = (self.indptr.size - 1,) * 2
        self._plans = {}

    @classmethod
    def from_scipy(cls, um):
        m = um.tocsr().copy()
        m.sort_indices()
        data = np.asarray(m.data)
        if not np.all(np.abs(data) == 1):
            raise ValueError('UM entries must be +1 / -1')
        return cls(m.indptr, m.indices, data.astype(np.int8))

    def tocsr(self):
        import scipy.sparse as sparse
        return sparse.csr_matrix((self.sign.astype(int), self.indices, self.indptr), shape=self.shape)

    def plan(self, ctx):
        p = self._plans.get(ctx.handle)
        if p is None or p.handle is None or p.ctx.handle is None:
            p = self._plans[ctx.handle] = ctx.route_plan(self.indptr, self.indices, self.sign)
        return p


def _as_um(um):
    return um if isinstance(um, UpstreamMatrix) else UpstreamMatrix.from_scipy(um)


def downstream(coords, flow_dir, settings):
    """Downstream cell id (1-based, -1 = outlet) of every cell (mrtm.py:85-120)."""
    return _hip.mrtm_downstream(coords, flow_dir, settings.ngridrow, settings.ngridcol)


def upstream(coords, dsid, settings):
    """[ncell, 9]: the 8 neighbour ids, inflowing ones first, and their count (mrtm.py:123-191)."""
    return _hip.mrtm_upstream(coords, dsid, settings.ngridrow, settings.ngridcol)


def upstream_genmatrix(upid):
    """UM = UP - I (mrtm.py:194-230)."""
    return UpstreamMatrix(*_hip.mrtm_um_csr(upid))


def n_substeps(nday, dt):
    return int(nday * 24 * 3600 / dt)      # mrtm.py:35


def route_series_device(ctx, um, nmonths, spinup_months, ndays, dt, d_flow_dist, d_velocity, d_area, d_runoff,
                        d_S0=None, d_chs=None, d_avg=None, d_S_end=None, d_F_end=None, flags=0, want_chs=True):
    """Device-resident variant: runoff and outputs stay in HBM. Returns (d_chs, d_avg)."""
    um = _as_um(um)
    ncell = um.shape[0]
    if d_avg is None:
        d_avg = ctx.empty((ncell, nmonths))
    if d_chs is None and want_chs:
        d_chs = ctx.empty((ncell, nmonths))
    ctx.route_series(um.plan(ctx), nmonths, spinup_months, ndays, dt, d_flow_dist, d_velocity, d_area, d_runoff,
                     d_S0, d_chs, d_avg, d_S_end, d_F_end, flags)
    return d_chs, d_avg


def route_series(UM, flow_dist, velocity, area, runoff, ndays, spinup_months, S0=None, dt=10800, device=0, flags=0,
                 prepare=True):
    """Spin-up then simulation over all months (components.py:273-294). Returns (ChStorage, Avg_ChFlow, F_end)."""
    ctx = _hip.get_context(device)
    um = _as_um(UM)
    runoff = np.asarray(runoff, dtype=np.float64)
    ncell, nmonths = runoff.shape
    if um.shape[0] != ncell:
        raise ValueError('UM is {} x {} but runoff has {} cells'.format(um.shape[0], um.shape[0], ncell))
    # which cells can fire follows from the lengths, velocities and dt this call holds: the prepared plan (folded leaves,
    # single running sums: xh_route_plan_prepare) is the one run_model() routes on -- a cheap no-op when nothing changed
    # (prepare=False: route on whatever plan there is -- the plan of pairs, unless somebody prepared it)
    if prepare and not flags & _hip.XH_ROUTE_EXACT:
        um.plan(ctx).prepare(flow_dist, velocity, dt)
    bufs = [ctx.upload(flow_dist), ctx.upload(velocity), ctx.upload(area), ctx.upload(runoff)]
    d_S0 = None if S0 is None else ctx.upload(S0)
    d_F = ctx.empty(ncell)
    d_chs, d_avg = route_series_device(ctx, um, nmonths, spinup_months, ndays, dt, *bufs, d_S0=d_S0, d_F_end=d_F,
                                       flags=flags)
    out = d_chs.download(), d_avg.download(), d_F.download()
    for b in bufs + [d_S0, d_F, d_chs, d_avg]:
        if b is not None:
            b.free()
    return out


def streamrouting(L, S0, F0, ChV, q, area, nday, dt, UM, device=0, flags=0):
    """One month of routing (mrtm.py:16-82). ``F0`` is accepted and ignored: the reference overwrites it (:50)."""
    ctx = _hip.get_context(device)
    um = _as_um(UM)
    ncell = um.shape[0]
    q = np.asarray(q, dtype=np.float64).reshape(ncell, 1)
    if not flags & _hip.XH_ROUTE_EXACT:
        um.plan(ctx).prepare(L, ChV, dt)      # (see route_series)
    bufs = [ctx.upload(L), ctx.upload(ChV), ctx.upload(area), ctx.upload(q), ctx.upload(S0)]
    d_avg, d_S, d_F = ctx.empty((ncell, 1)), ctx.empty(ncell), ctx.empty(ncell)
    ctx.route_series(um.plan(ctx), 1, 0, [int(nday)], dt, bufs[0], bufs[1], bufs[2], bufs[3], bufs[4], None, d_avg,
                     d_S, d_F, flags)
    out = d_S.download(), d_avg.download().reshape(ncell), d_F.download()
    for b in bufs + [d_avg, d_S, d_F]:
        b.free()
    return out
