"""The committed test world of the single-sum routing form's hard corner.

Reference: xanthos/routing/mrtm.py:54-69.  A cell that does not fire in the trial step (S1 >= 0, :54) is updated with the
ADJUSTED inflows (:66-69).  If an upstream neighbour fired, its adjusted outflow is smaller than its trial outflow, and the
cell's storage can come out NEGATIVE (S1 >= 0 > S2).  Its outflow turns negative in the next sub-step (:50), and cells
downstream fire that cannot by construction (velocity * dt / length < 1).  It takes a cell that can fire with an upstream
neighbour that can: site k below is

    u_0 -> ... -> u_{n_up-1} -> B -> A -> c_1 -> c_2 -> ... -> c_{n_below} -> (next site's c_3, or the outlet)

with velocity * dt / length = ratio_A at A (17.8 at site 0: cell 29431 of the synthetic 67,420-cell world, where round 5's
experiment went wrong), ratio_B at B, 0.27 elsewhere, three-cell tributaries joining c_3, c_6, c_9, c_12, and a capable cell
INSIDE the halo of site 1 (the halo has to start again below it).  n_up decides whether the corner occurs (A's lateral
inflow has to sit between 13 % and 23 % of what reaches B per sub-step, and the two cells have to fire in phase), so the
seeds are picked by looking: 3 (site 0: 59 negative storages, site 2: 7; negative flows four cells down; six cells fire that
cannot by construction), 13 and 20 (site 1: 56).  `instrumented` is the reference's sub-step loop
(a restatement for this test only) counting fired cells, negative storages and negative flows, so that the test can assert
that the corner it is about actually happens.
"""
import numpy as np


def make(seed=3, dt=10800.0):
    rng = np.random.default_rng(seed)
    ds, role = {}, {}
    sites = [dict(n_up=4, ratio_a=17.8, ratio_b=3.6, n_below=14), dict(n_up=6, ratio_a=8.0, ratio_b=2.2, n_below=14),
             dict(n_up=5, ratio_a=2.9, ratio_b=19.5, n_below=6)]
    prev_join = None
    for k, s in enumerate(sites):
        names = ['s%d_u%d' % (k, i) for i in range(s['n_up'])] + ['s%d_B' % k, 's%d_A' % k] + \
                ['s%d_c%d' % (k, i) for i in range(1, s['n_below'] + 1)]
        for a, b in zip(names[:-1], names[1:]):
            ds[a] = b
        ds[names[-1]] = prev_join                       # site k drains into site k - 1's chain (None: the outlet)
        prev_join = 's%d_c3' % k
        for j, at in enumerate((3, 6, 9, 12)):
            if at > s['n_below']:
                continue
            t = ['s%d_t%d_%d' % (k, j, i) for i in range(3)]
            for a, b in zip(t[:-1], t[1:]):
                ds[a] = b
            ds[t[-1]] = 's%d_c%d' % (k, at)
        role['s%d_A' % k] = s['ratio_a']
        role['s%d_B' % k] = s['ratio_b']
    role['s1_c5'] = 1.8                                 # a capable cell inside site 1's halo
    role['s2_c6'] = 4.0                                 # ... and a capable cell with a capable neighbour just above the outlet of site 2
    role['s2_c5'] = 1.3
    cells = sorted(ds)
    perm = rng.permutation(len(cells))
    idx = {c: int(perm[i]) for i, c in enumerate(cells)}
    n = len(cells)
    rows = [[(i, -1)] for i in range(n)]
    for c, d in ds.items():
        if d is not None:
            rows[idx[d]].append((idx[c], 1))
    indptr, indices, data = [0], [], []
    for r in rows:
        for col, sg in sorted(r):
            indices.append(col)
            data.append(sg)
        indptr.append(len(indices))
    L = np.full(n, 40e3)
    v = np.full(n, 1.0)
    area = rng.uniform(2450.0, 2550.0, n)
    for c, ratio in role.items():
        L[idx[c]] = dt * v[idx[c]] / ratio
    return idx, (np.array(indptr), np.array(indices), np.array(data)), L, v, area


def runoff(n, nmonths=6, seed=3):
    rng = np.random.default_rng(seed)
    q = rng.gamma(2.0, 30.0, (n, nmonths))
    q[:, 2] *= 0.02                                     # a dry month: negative flows travel further
    return q


def instrumented(csr, L, v, area, q, ndays, dt=10800.0):
    """mrtm.py:16-82 month by month from zero storage; returns (fired, negative storages, negative trial flows) per cell."""
    import scipy.sparse as sparse
    indptr, indices, data = csr
    n = len(L)
    um = sparse.csr_matrix((np.asarray(data, dtype=float), indices, indptr), shape=(n, n))
    tauinv = v / L
    S = np.zeros(n)
    fired, neg_s, neg_f = np.zeros(n, int), np.zeros(n, int), np.zeros(n, int)
    for m in range(q.shape[1]):
        nt = int(ndays[m] * 86400 / dt)
        erl = q[:, m] * area * 1000.0 / (ndays[m] * 86400)
        for _ in range(nt):
            F = S * tauinv
            neg_f += F < 0
            dsdt = um.dot(F) + erl
            sx = dsdt * dt < -S
            if sx.any():
                fired += sx
                F[sx] = dsdt[sx] + F[sx] + S[sx] / dt
                S[sx] = 0
                keep = ~sx
                dsdt[keep] = um.dot(F)[keep] + erl[keep]
                S[keep] += dsdt[keep] * dt
            else:
                S += dsdt * dt
            neg_s += S < 0
    return fired, neg_s, neg_f
