"""Which cells of the synthetic world ever fire (mrtm.py:54-63: storage driven negative, outflow capped) and how far
downstream of the cells that can fire BY CONSTRUCTION (velocity * dt / length > 1 - 2^-20) they sit.

CPU experiment behind the routing plan's static "plain set" (DESIGN.md 4.3): the plain form of a dataflow unit is valid
when no upstream neighbour of its cells fires; cells below a firing cell can be driven negative by the ADJUSTED inflow
although their own ratio is < 1, so the plan closes the set over a few downstream levels.  Uses the oracle only.

    python tools/fired_cells.py [months] [procs]
"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import abcd as o_abcd, months as o_months, mrtm as o_mrtm, pm as o_pm      # noqa: E402
from xanthos_amd import synth                                                           # noqa: E402


def fired_in_group(job):
    um, cells, L, V, A, q, ndays, spin, dt = job
    sub = um[cells][:, cells].tocsr()
    sub.sort_indices()
    L, V, A, q = L[cells], V[cells], A[cells], np.nan_to_num(q[cells])
    tauinv, dtinv = V / L, 1.0 / dt
    S = np.zeros(len(cells))
    fired = np.zeros(len(cells), dtype=np.int64)
    for nm in list(range(spin)) + list(range(q.shape[1])):
        nt = int(ndays[nm] * 24 * 3600 / dt)
        erl = (q[:, nm] * A) * 1000.0 / (ndays[nm] * 24 * 3600)
        for _ in range(nt):
            F = S * tauinv
            dsdt = sub.dot(F) + erl
            sx = (dsdt * dt) < (-S)
            if sx.any():
                fired += sx
                F[sx] = dsdt[sx] + F[sx] + S[sx] * dtinv
                S[sx] = 0
                keep = ~sx
                dsdt[keep] = (sub.dot(F))[keep] + erl[keep]
                S[keep] += dsdt[keep] * dt
            else:
                S += dsdt * dt
    return cells, fired


def main():
    nm = int(sys.argv[1]) if len(sys.argv) > 1 else 120
    procs = int(sys.argv[2]) if len(sys.argv) > 2 else os.cpu_count()
    w = synth.make_world()
    from types import SimpleNamespace as NS
    st = NS(ngridrow=w.nrow, ngridcol=w.ncol)
    ds = o_mrtm.downstream(w.coords, w.flow_dir, st)
    um = o_mrtm.upstream_genmatrix(o_mrtm.upstream(w.coords, ds, st)).tocsr()
    t = time.time()
    f = synth.make_forcing(w, nm)
    pet = o_pm.run_pmpet(synth.data_bag(w, f), w.ncell, w.nlcs, 1961, 1961 + nm // 12 - 1, 0, 6, w.lc_years)
    _, _, q, _ = o_abcd.abcd_execute(w.n_basins, w.basin_ids, pet, f['precip'], f['abcd_tmin'], w.abcd_pars, nm,
                                     min(120, nm), -1)
    print('runoff of {} months in {:.0f} s'.format(nm, time.time() - t), flush=True)
    ndays = o_months.set_month_arrays(nm, 1961, 1961 + nm // 12 - 1)[:, 2]
    dt = 10800.0
    groups = o_mrtm.network_groups(um, procs * 2)
    jobs = [(um, c, w.flow_dist, w.velocity, w.area, q, ndays, min(120, nm), dt) for c in groups]
    import multiprocessing as mp
    t = time.time()
    with mp.get_context('fork').Pool(procs) as pool:
        parts = pool.map(fired_in_group, jobs, chunksize=1)
    fired = np.zeros(w.ncell, dtype=np.int64)
    for cells, fr in parts:
        fired[cells] = fr
    print('routed in {:.0f} s'.format(time.time() - t))
    cap = ~((w.velocity / w.flow_dist) * dt <= 1.0 - 1.0 / 1048576.0)
    did = fired > 0
    print('cells: {}   can fire by construction: {}   fired: {}   fired without being in the set: {}'.format(
        w.ncell, int(cap.sum()), int(did.sum()), int((did & ~cap).sum())))
    # distance (edges) of each unexpected cell below the nearest cell of the set
    dsi = np.where(ds > 0, ds - 1, -1)
    level = np.full(w.ncell, -1)
    level[cap] = 0
    frontier = np.nonzero(cap)[0]
    for lv in range(1, 64):
        nxt = dsi[frontier]
        nxt = np.unique(nxt[nxt >= 0])
        nxt = nxt[level[nxt] < 0]
        if not len(nxt):
            break
        level[nxt] = lv
        frontier = nxt
    un = did & ~cap
    hist = np.bincount(level[un][level[un] >= 0], minlength=1)
    print('levels below the set of the unexpected cells (1 = direct downstream neighbour):', hist.tolist(),
          ' not below any:', int((level[un] < 0).sum()))
    for lv in range(0, 6):
        closed = (level >= 0) & (level <= lv)
        need_pairs = np.zeros(w.ncell, dtype=bool)
        src = np.nonzero(closed)[0]
        tgt = dsi[src]
        need_pairs[tgt[tgt >= 0]] = True
        print('closure over {} level(s): {} cells in the set, {} cells need pairs, {} unexpected cells left'.format(
            lv, int(closed.sum()), int(need_pairs.sum()), int((did & ~closed).sum())))
    np.save('/tmp/fired_cells.npy', fired)


if __name__ == '__main__':
    main()
