"""Differential fuzz of the REASSOCIATED routing form (k_mrtm_rsum) against the numpy oracle (run on the GPU box).

tools/fuzz_routing.py's generator -- random tree worlds of random size, month counts / spin-ups / time steps / initial storage,
NaN runoff cells, stagnant channels and channels shorter than velocity x dt --, every case routed with XH_ROUTE_REASSOC and held
to the form's bar: identical NaN masks, |x - ref| <= 1e-9 |ref| + 1e-3 m3 (storage) / 1e-9 m3/s (flows) -- twice: on the plain
reassociated plan (pairs of sums), and again after xh_route_plan_prepare (leaves that cannot fire folded into their downstream
cells' lanes; single running sums, the cells that may fire next to one that may in pair units with a halo below them) -- a case
whose data break the prepared plan's assumptions (negative initial storage or runoff, a halo of zero cells below a corner cell)
must trip the guard and still come out right.
Prints the worst relative error seen and how the prepared plans fared.  Usage: python tools/fuzz_reassoc.py [n_cases] [seed]
"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
from oracle import mrtm as o_mrtm                         # noqa: E402
from xanthos_amd import _hip                              # noqa: E402
from xanthos_amd.routing import mrtm                      # noqa: E402
import importlib.util                                     # noqa: E402

spec = importlib.util.spec_from_file_location('fuzz_routing', os.path.join(os.path.dirname(os.path.abspath(__file__)), 'fuzz_routing.py'))
fz = importlib.util.module_from_spec(spec)
spec.loader.exec_module(fz)


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 100
    rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 5150)
    t0, worst, kernels = time.time(), 0.0, {}
    folds = {'folded': 0, 'tripped': 0, 'leaves': 0, 'single': 0, 'pair_cells': 0, 'halo0': 0, 'halo0_tripped': 0}
    for k in range(n):
        c = fz.gen_case(rng)
        # data that test the folded leaves' guard: negative initial storage / runoff negative by a rounding error of a runoff
        # model (must pass without a trip) / plainly negative runoff (must trip wherever a folded leaf meets it)
        if k % 7 == 3:
            c.S0 = rng.uniform(-1e5, 1e7, c.w.ncell)
        elif k % 7 == 5:
            c.q[rng.random(c.q.shape) < 0.02] = -1e-14
        elif k % 7 == 6:
            c.q[rng.random(c.q.shape) < 0.01] = -5.0
        ref = o_mrtm.route_series(c.um.tocsr(), c.L, c.v, c.w.area, c.q, c.ndays, c.spin, S0=c.S0, dt=c.dt)
        ctx = _hip.get_context(0)
        case_worst, kern, tag = 0.0, None, ''
        halo0 = k % 7 == 2            # no halo below the cells that may leave negative storage: trips wherever that happens
        for prepared in (False, True):
            if prepared:
                if halo0:
                    os.environ['XH_RSUM_HALO'] = '0'
                c.um.plan(ctx).prepare(c.L, c.v, c.dt)
                os.environ.pop('XH_RSUM_HALO', None)
            got = mrtm.route_series(c.um, c.L, c.v, c.w.area, c.q, c.ndays, c.spin, S0=c.S0, dt=c.dt, flags=_hip.XH_ROUTE_REASSOC,
                                    prepare=False)
            if not prepared:
                kern = c.um.plan(ctx).info()['last_tree_kernel']
                kernels[kern] = kernels.get(kern, 0) + 1
            else:
                ri = c.um.plan(ctx).rsum_info()
                folds['folded'] += int(ri['folded'] > 0)
                folds['tripped'] += int(ri['fold_disabled'] > 0)
                folds['leaves'] += int(ri['folded'])
                folds['single'] += int(ri['pair_cells'] >= 0)
                folds['pair_cells'] += max(int(ri['pair_cells']), 0)
                folds['halo0'] += int(halo0)
                folds['halo0_tripped'] += int(halo0 and ri['fold_disabled'] > 0)
                tag = 'folded {:4d} pair cells {:4d}{}{}'.format(int(ri['prepared_folded']), int(ri['prepared_pair_cells']),
                                                                ' (guard tripped)' if ri['fold_disabled'] else '', ' halo 0' if halo0 else '')
            for x, r, atol in zip(got, ref, (1e-3, 1e-9, 1e-9)):
                if not np.array_equal(np.isnan(x), np.isnan(r)):
                    raise AssertionError('case {} (prepared {}): NaN masks differ ({} cells, {} months, dt {})'.format(k, prepared, c.ncell, c.nm, c.dt))
                m = ~np.isnan(r)
                err = np.abs(x[m] - r[m])
                if not (err <= 1e-9 * np.abs(r[m]) + atol).all():
                    raise AssertionError('case {} (prepared {}): {} values beyond the bar, worst {:.3e} ({} cells, {} months, spin {}, dt {})'.format(
                        k, prepared, int((err > 1e-9 * np.abs(r[m]) + atol).sum()), float(err.max()), c.ncell, c.nm, c.spin, c.dt))
                big = np.abs(r[m]) > 1e6 * atol
                if big.any():
                    case_worst = max(case_worst, float((err[big] / np.abs(r[m][big])).max()))
        worst = max(worst, case_worst)
        print('case {:3d}: {:5d} cells {:2d} months spin {:2d} dt {:6.0f} kernel {} worst rel {:.2e}  {}'.format(
            k, c.ncell, c.nm, c.spin, c.dt, kern, case_worst, tag), flush=True)
    print('{} cases within 1e-9 in {:.0f} s; worst relative error {:.2e}; kernels used {} (4 = k_mrtm_rsum; months shorter than the '
          'lane lags fall to the bit-exact lock-step kernel); prepared plans: {} routed with folded leaves ({} leaves), {} on single sums '
          '({} cells in pair units), {} gave the prepared plan up on a guard trip ({} of the {} cases without a halo)'.format(
              n, time.time() - t0, worst, kernels, folds['folded'], folds['leaves'], folds['single'], folds['pair_cells'],
              folds['tripped'], folds['halo0_tripped'], folds['halo0']))


if __name__ == '__main__':
    main()
