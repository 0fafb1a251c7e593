"""Differential fuzz of the routing kernels against the numpy oracle (run on the GPU box).

Random tree worlds of random size, random month counts / spin-ups / time steps / initial storage, NaN runoff cells and
channels shorter than velocity x dt; every dataflow variant (time-skewed, lock-step, workgroup per network) must match
the oracle bit for bit.  Usage: python tools/fuzz_routing.py [n_cases] [seed]
"""
import os
import sys
import time
from types import SimpleNamespace as NS

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
EXACT = 256      # XH_ROUTE_EXACT: this tool holds the BIT-EXACT kernels to the oracle's bits (the default form is tested to 1e-9 elsewhere)
from oracle import mrtm as o_mrtm            # noqa: E402
from xanthos_amd import _hip, synth           # noqa: E402
from xanthos_amd.routing import mrtm          # noqa: E402


def gen_case(rng):
    ncell = int(rng.choice([12, 70, 300, 900, 2500, 6000]))
    nrow = int(np.ceil(np.sqrt(ncell * 2.2))) + 6
    ncol = 2 * nrow
    w = synth.make_world(nrow=nrow, ncol=ncol, ncell=ncell, n_basins=int(rng.integers(1, 8)), seed=int(rng.integers(1, 1 << 30)),
                         outlet_frac=float(rng.choice([0.002, 0.02, 0.1])))
    st = NS(ngridrow=w.nrow, ngridcol=w.ncol)
    um = mrtm.upstream_genmatrix(mrtm.upstream(w.coords, mrtm.downstream(w.coords, w.flow_dir, st), st))
    nm = int(rng.integers(1, 15))
    spin = int(rng.integers(0, nm + 1))
    dt = float(rng.choice([10800, 10800, 7200, 21600, 17280, 43200]))
    ndays = rng.choice([28, 29, 30, 31], nm)
    q = rng.gamma(2.0, 30.0, (w.ncell, nm))
    if rng.random() < 0.5:
        q[rng.random(w.ncell) < 0.01] = np.nan
    L = w.flow_dist.copy()
    L[rng.random(w.ncell) < 0.03] = 3e3                       # cells that fire most sub-steps
    v = w.velocity.copy()
    if rng.random() < 0.3:
        v[rng.random(w.ncell) < 0.01] = 0.0                    # stagnant channels
    S0 = rng.uniform(0, 1e7, w.ncell) if rng.random() < 0.5 else None
    return NS(w=w, um=um, nm=nm, spin=spin, dt=dt, ndays=ndays, q=q, L=L, v=v, S0=S0, ncell=ncell)


def one_case(rng, k):
    c = gen_case(rng)
    w, um, nm, spin, dt, ndays, q, L, v, S0, ncell = c.w, c.um, c.nm, c.spin, c.dt, c.ndays, c.q, c.L, c.v, c.S0, c.ncell
    ref = o_mrtm.route_series(um.tocsr(), L, v, w.area, q, ndays, spin, S0=S0, dt=dt)
    used = []
    for flags in (0, 8, 4, 0):
        got = mrtm.route_series(um, L, v, w.area, q, ndays, spin, S0=S0, dt=dt, flags=flags | EXACT)
        for a, b in zip(got, ref):
            if not np.array_equal(a, b, equal_nan=True):
                plan = um.plan(_hip.get_context(0))
                bad = ~((a == b) | (np.isnan(a) & np.isnan(b)))
                raise AssertionError('case {} flags {}: ncell {} months {} spin {} dt {} mismatch in {} values of {} cells; '
                                     'plan {}'.format(k, flags, ncell, nm, spin, dt, int(bad.sum()),
                                                      int(bad.reshape(len(bad), -1).any(axis=1).sum()), plan.info()))
        used.append(um.plan(_hip.get_context(0)).info()['last_tree_kernel'])
    adaptive = 0
    return ncell, nm, spin, dt, used, adaptive


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 40
    rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 2024)
    t0 = time.time()
    kernels = {}
    n_adaptive = 0
    for k in range(n):
        ncell, nm, spin, dt, used, adaptive = one_case(rng, k)
        kernels[used[0]] = kernels.get(used[0], 0) + 1
        n_adaptive += adaptive > 0
        print('case {:3d}: {:5d} cells {:2d} months spin {:2d} dt {:6.0f} kernels {} ok'.format(k, ncell, nm, spin, dt, used), flush=True)
    print('{} cases bit-exact in {:.0f} s; the time-skewed path (flags 0) used kernels {}'.format(n, time.time() - t0, kernels))


if __name__ == '__main__':
    main()
