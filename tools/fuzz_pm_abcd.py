"""Parity fuzz of the PM and ABCD kernels against the numpy oracle on random worlds and forcing (run on the GPU box).

Reports, per case, the worst |gpu - oracle| as a fraction of the north-star gate 1e-6 |ref| + 1e-9; fails above 1.
Forcing includes zeros produced by nan_to_num (missing relative humidity etc.), saturated air, extreme cold, and
NaN precipitation.  Usage: python tools/fuzz_pm_abcd.py [n_cases] [seed]
"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
from oracle import abcd as o_abcd, pm as o_pm                    # noqa: E402
from xanthos_amd import synth                                    # noqa: E402
from xanthos_amd.pet import penman_monteith as pm                # noqa: E402
from xanthos_amd.runoff import abcd                              # noqa: E402


def gate(x, ref):
    x, ref = np.asarray(x), np.asarray(ref)
    if not np.array_equal(np.isnan(x), np.isnan(ref)):
        return np.inf
    m = ~np.isnan(ref)
    return float(np.max(np.abs(x[m] - ref[m]) / (1e-6 * np.abs(ref[m]) + 1e-9))) if m.any() else 0.0


def one_case(rng, k):
    ncell = int(rng.choice([64, 500, 2000]))
    nrow = int(np.ceil(np.sqrt(ncell * 2.2))) + 6
    w = synth.make_world(nrow=nrow, ncol=2 * nrow, ncell=ncell, n_basins=int(rng.integers(1, 9)), seed=int(rng.integers(1, 1 << 30)))
    years = int(rng.integers(3, 7))
    nm = 12 * years
    y0 = int(rng.choice([1971, 1989, 1996, 2003]))
    f = synth.make_forcing(w, nm, seed=int(rng.integers(1, 1 << 30)))
    # hostile values
    for key in ('rhs', 'wind', 'rsds', 'rlds', 'tas', 'tmin'):
        f[key][rng.random(f[key].shape) < 0.002] = 0.0                       # what nan_to_num makes of a missing value
    f['rhs'][rng.random(f['rhs'].shape) < 0.01] = 100.0
    f['rhs'][rng.random(f['rhs'].shape) < 0.01] = 99.99995
    f['tas'][rng.random(f['tas'].shape) < 0.005] -= 60.0
    d = synth.data_bag(w, f)
    pet = pm.run_pmpet(d, w.ncell, w.nlcs, y0, y0 + years - 1, 0, 6, w.lc_years)
    r_pet = o_pm.run_pmpet(d, w.ncell, w.nlcs, y0, y0 + years - 1, 0, 6, w.lc_years)
    spin = int(rng.integers(25, nm + 1))
    tmin = f['abcd_tmin'] if rng.random() < 0.8 else None
    got = abcd.abcd_execute(w.n_basins, w.basin_ids, r_pet, f['precip'], tmin, w.abcd_pars, nm, spin, -1)
    ref = o_abcd.abcd_execute(w.n_basins, w.basin_ids, r_pet, f['precip'], tmin, w.abcd_pars, nm, spin, 1)
    g = [gate(pet, r_pet)] + [gate(a, b) for a, b in zip(got[1:], ref[1:])]
    if max(g) > 1.0:
        raise AssertionError('case {}: gate fractions {} (ncell {} months {} spin {})'.format(k, g, ncell, nm, spin))
    return ncell, nm, spin, g


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 20
    rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 2025)
    t0 = time.time()
    worst = np.zeros(4)
    for k in range(n):
        ncell, nm, spin, g = one_case(rng, k)
        worst = np.maximum(worst, g)
        print('case {:3d}: {:5d} cells {:3d} months spin {:3d}  gate fractions pet {:.1e} aet {:.1e} q {:.1e} sav {:.1e}'.format(
            k, ncell, nm, spin, *g), flush=True)
    print('{} cases within tolerance in {:.0f} s; worst fractions of the 1e-6 gate: pet {:.1e} aet {:.1e} q {:.1e} sav {:.1e}'.format(
        n, time.time() - t0, *worst))


if __name__ == '__main__':
    main()
