"""Modified River Transport Model (oracle; test infrastructure only).

CPU numpy / scipy.sparse restatement of xanthos/routing/mrtm.py and of the
month loops that drive it in xanthos/components.py:249-296.

Reference map:
  D8 decode -> downstream cell id ........ downstream, make_flowdirgrid   mrtm.py:85-120, 233-258
  8-neighbour scan -> upstream table ..... upstream                       mrtm.py:123-191
  UM = UP - I (CSR, int) ................. upstream_genmatrix             mrtm.py:194-230
  one month of 3-hourly explicit Euler ... streamrouting                  mrtm.py:16-82
  spin-up + simulation month loops ....... Components.calculate_routing   components.py:249-296

Third-party arithmetic: ``scipy.sparse`` CSR mat-vec (setup pin scipy>=1.6, here 1.15.x): row i of ``UM.dot(F)``
is ``0 + sum_j data[j]*F[col[j]]`` accumulated in stored (ascending column) order.
"""
import numpy as np
import scipy.sparse as sparse

# bit groups of the D8 code that move the target one column right / left, one row up / down (mrtm.py:236-240)
_RIGHT = 1 | 2 | 128
_LEFT = 8 | 16 | 32
_UP = 32 | 64 | 128
_DOWN = 2 | 4 | 8


def _grid_of_ids(coords, nrow, ncol):
    grid = np.zeros((nrow, ncol), dtype=np.int64)
    grid[coords[:, 4].astype(int) - 1, coords[:, 3].astype(int) - 1] = coords[:, 0].astype(np.int64)
    return grid


def downstream(coords, flowdir, settings):
    """1-based id of the cell each cell drains to, -1 for outlets (mrtm.py:85-120)."""
    nrow, ncol = settings.ngridrow, settings.ngridcol
    grid = _grid_of_ids(coords, nrow, ncol)
    r0 = coords[:, 4].astype(int) - 1
    c0 = coords[:, 3].astype(int) - 1

    code = np.where(flowdir == -9999., 0, flowdir).astype(int)          # mrtm.py:243-245
    dr = np.zeros(len(code), dtype=int)
    dc = np.zeros(len(code), dtype=int)
    dr[(code & _DOWN) != 0] = -1
    dr[(code & _UP) != 0] = 1
    dc[(code & _RIGHT) != 0] = 1
    dc[(code & _LEFT) != 0] = -1
    r = r0 + dr
    c = c0 + dc

    off = (c < 0) | (c > ncol - 1)                                      # longitude wrap quirk, mrtm.py:100-102
    c[off] = np.mod(c[off] + 1, ncol)
    off = (r < 0) | (r > nrow - 1)                                      # off the top/bottom -> self, mrtm.py:104-106
    r[off] = r0[off]
    c[off] = c0[off]

    ds = grid[r, c]
    ds[(ds == 0) | (ds == coords[:, 0])] = -1                           # ocean or self -> outlet, mrtm.py:116-118
    return ds


def upstream(coords, dsid, settings):
    """[ncell, 9] table: neighbour ids (inflowing first) + inflow count (mrtm.py:123-191)."""
    nrow, ncol = settings.ngridrow, settings.ngridcol
    grid = _grid_of_ids(coords, nrow, ncol)
    n = coords.shape[0]
    r0 = coords[:, 4].astype(int) - 1
    c0 = coords[:, 3].astype(int) - 1
    ids = coords[:, 0].astype(int)

    nbr = np.zeros((n, 8), dtype=int)
    flows_in = np.zeros((n, 8), dtype=bool)
    k = 0
    for dr in (-1, 0, 1):
        for dc in (-1, 0, 1):
            if dr == 0 and dc == 0:
                continue
            # reference neighbour order: (-1,-1),(-1,0),(-1,1),(0,-1),(0,1),(1,-1),(1,0),(1,1)
            r = r0 + dr
            c = c0 + dc
            ok = (r >= 0) & (c >= 0) & (r <= nrow - 1) & (c <= ncol - 1)   # no wrap here, mrtm.py:150
            nbr[ok, k] = grid[r[ok], c[ok]]
            land = nbr[:, k] != 0
            flows_in[land, k] = dsid[nbr[land, k] - 1] == ids[land]
            k += 1

    order = np.argsort(~flows_in, axis=1)          # default (quicksort) like the reference, mrtm.py:172
    sorted_nbr = np.take_along_axis(nbr, order, axis=1)
    count = flows_in.sum(axis=1)
    return np.concatenate((sorted_nbr, count[:, None]), axis=1)


def upstream_genmatrix(upid):
    """UM = UP - I as scipy CSR (mrtm.py:194-230)."""
    n = upid.shape[0]
    cnt = upid[:, 8]
    rows = np.repeat(np.arange(n), cnt)
    cols = np.concatenate([upid[i, :cnt[i]] for i in range(n)] + [np.zeros(0, dtype=int)]).astype(int) - 1
    up = sparse.coo_matrix((np.ones(len(rows), dtype=int), (rows, cols)), shape=(n, n))
    return up - sparse.eye(n, dtype=int)


def streamrouting(L, S0, F0, ChV, q, area, nday, dt, UM):
    """One month of channel routing (mrtm.py:16-82). Returns (S, Favg, F)."""
    nt = int(nday * 24 * 3600 / dt)
    S = np.copy(S0)
    F = np.copy(F0)
    favg = np.zeros(L.shape[0], dtype=float)
    tauinv = ChV / L
    dtinv = 1. / dt
    erlateral = (q * area) * (1e6 / 1e3) / (nday * 24 * 3600)

    for _ in range(nt):
        F = S * tauinv
        dsdt = UM.dot(F) + erlateral
        sx = (dsdt * dt) < (-S)
        if sx.any():
            F[sx] = dsdt[sx] + F[sx] + S[sx] * dtinv
            S[sx] = 0
            keep = ~sx
            dsdt[keep] = (UM.dot(F))[keep] + erlateral[keep]
            S[keep] += dsdt[keep] * dt
        else:
            S += dsdt * dt
        favg += F
    favg /= nt
    return S, favg, F


def streamrouting_fused(L, S0, F0, ChV, q, area, nday, dt, UM):
    """One month in the arithmetic of the REASSOCIATED kernel form (k_mrtm_rsum, csrc/xh_mrtm_wave_unit.h RSUM; test
    infrastructure like the rest of this package): the same explicit Euler step and "excess flow" rule as ``streamrouting``
    (mrtm.py:50-69), restated so that a cell needs only the SUM of its upstream neighbours' flows -- in any order -- and
    eight operations:

        base = S a + erl dt                    a = 1 - (ChV / L) dt
        S1   = base + (sum F) dt               the trial storage; Sx of mrtm.py:54 (dSdt dt < -S) is S1 < 0
        F2   = F + min(S1, 0) / dt             mrtm.py:60: inbound + lateral + S / dt = F + S1 / dt
        S    = Sx ? 0 : base + (sum F2) dt     mrtm.py:63, 69

    numpy has no fma, so this differs from the kernel by roundings of its own; both are held to 1e-9 of ``streamrouting``
    (tests/test_oracle_golden.py here, tests/test_gpu_reassoc.py on the device; measured ~1e-12)."""
    nt = int(nday * 24 * 3600 / dt)
    n = L.shape[0]
    up = (UM + sparse.eye(n, dtype=int)).tocsr().astype(float)          # UP = UM + I: inflow only
    tauinv = ChV / L
    a = 1.0 - tauinv * dt
    dtinv = 1.0 / dt
    erldt = ((q * area) * (1e6 / 1e3) / (nday * 24 * 3600)) * dt
    S = np.copy(S0)
    F = np.copy(F0)
    favg = np.zeros(n, dtype=float)
    with np.errstate(invalid='ignore'):
        for _ in range(nt):
            F0_ = S * tauinv
            base = S * a + erldt
            S1 = up.dot(F0_) * dt + base
            sx = S1 < 0.0
            F = F0_ + np.where(sx, S1, 0.0) * dtinv
            S2 = up.dot(F) * dt + base
            S = np.where(sx, 0.0, S2)
            favg += F
    favg /= nt
    return S, favg, F


def route_series(UM, flow_dist, velocity, area, runoff, ndays, spinup_months, S0=None, dt=10800):
    """Month loops of Components.calculate_routing (components.py:273-294).

    ``runoff`` [ncell, nmonths]; ``ndays`` [nmonths].  Spin-up routes months 0..spinup_months-1, then the
    simulation restarts at month 0 from the spin-up end state.  Returns (ChStorage, Avg_ChFlow, F_end).
    """
    ncell, nmonths = runoff.shape
    chs = np.zeros((ncell, nmonths))
    avg = np.zeros((ncell, nmonths))
    S = np.zeros(ncell) if S0 is None else np.asarray(S0, dtype=float).copy()
    F = np.zeros(ncell)
    for nm in list(range(spinup_months)) + list(range(nmonths)):
        S, favg, F = streamrouting(flow_dist, S, F, velocity, runoff[:, nm], area, ndays[nm], dt, UM)
        chs[:, nm] = S
        avg[:, nm] = favg
    return chs, avg, F


# ---------------------------------------------------------------------------------------------------------------------
# The same month loops, river network by river network (checker / CPU baseline at full size; test infrastructure).
#
# Rows of UM only couple cells of one river network, so routing a union of whole networks on its own gives every cell
# of it the bits of the whole-grid run: a row's sum is taken over the same terms in the same stored (ascending column)
# order -- `network_groups` keeps each group's cells ascending, so the re-indexed columns keep their order -- and the
# reference's global ``sx.any()`` branch (mrtm.py:56-76) recomputes ``UM.dot(F)`` only to the same bits for rows none of
# whose terms changed.  ``tests/test_oracle_golden.py`` holds this to the serial loops bit for bit.


def network_groups(UM, n_groups):
    """Cells of whole river networks, dealt largest-first onto ``n_groups`` groups (each sorted ascending)."""
    from scipy.sparse.csgraph import connected_components
    ncomp, label = connected_components(UM, directed=False)
    sizes = np.bincount(label, minlength=ncomp)
    load = np.zeros(n_groups, dtype=np.int64)
    owner = np.empty(ncomp, dtype=np.int64)
    for comp in np.argsort(-sizes, kind='stable'):
        g = int(np.argmin(load))
        owner[comp] = g
        load[g] += sizes[comp]
    cell_group = owner[label]
    return [np.nonzero(cell_group == g)[0] for g in range(n_groups) if load[g] > 0]


def _route_group(job):
    import time
    UM, cells, flow_dist, velocity, area, runoff, ndays, spinup_months, S0, dt = job
    sub = UM[cells][:, cells].tocsr()
    sub.sort_indices()
    t = time.process_time()
    chs, avg, F = route_series(sub, flow_dist[cells], velocity[cells], area[cells], np.ascontiguousarray(runoff[cells]),
                               ndays, spinup_months, None if S0 is None else np.asarray(S0)[cells], dt)
    return cells, chs, avg, F, time.process_time() - t


def route_series_by_network(UM, flow_dist, velocity, area, runoff, ndays, spinup_months, S0=None, dt=10800,
                            n_procs=1):
    """``route_series`` over disjoint groups of whole river networks, ``n_procs`` worker processes.

    Returns (ChStorage, Avg_ChFlow, F_end, cpu_seconds) -- ``cpu_seconds`` is the process time the workers spent in the
    month loops, summed: the one-thread cost of the run (the reference routes on one thread)."""
    UM = UM.tocsr()
    ncell, nmonths = runoff.shape
    groups = network_groups(UM, max(1, int(n_procs)))
    jobs = [(UM, c, flow_dist, velocity, area, runoff, ndays, spinup_months, S0, dt) for c in groups]
    if len(jobs) > 1 and n_procs > 1:
        import multiprocessing as mp
        with mp.get_context('fork').Pool(min(int(n_procs), len(jobs))) as pool:      # fork: the arrays are not copied
            parts = pool.map(_route_group, jobs, chunksize=1)
    else:
        parts = [_route_group(j) for j in jobs]
    chs, avg, F = np.zeros((ncell, nmonths)), np.zeros((ncell, nmonths)), np.zeros(ncell)
    cpu = 0.0
    for cells, c, a, f, t in parts:
        chs[cells], avg[cells], F[cells] = c, a, f
        cpu += t
    return chs, avg, F, cpu
