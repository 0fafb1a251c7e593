"""Differential fuzz of the REASSOCIATED routing form (k_mrtm_rsum) against the numpy oracle (run on the GPU box).

tools/fuzz_routing.py's generator -- random tree worlds of random size, month counts / spin-ups / time steps / initial storage,
NaN runoff cells, stagnant channels and channels shorter than velocity x dt --, every case routed with XH_ROUTE_REASSOC and held
to the form's bar: identical NaN masks, |x - ref| <= 1e-9 |ref| + 1e-3 m3 (storage) / 1e-9 m3/s (flows) -- twice: on the plain
reassociated plan, and again after xh_route_plan_prepare (leaves that cannot fire folded into their downstream cells' lanes; a
case whose data break the fold's assumption -- negative initial storage, say -- must trip the guard and still come out right).
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
os.environ.pop('XH_ROUTE_REASSOC', None)             # (fuzz_routing pins the bit-exact form for its own runs; the library has not read it yet)


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 100
    rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 5150)
    t0, worst, kernels = time.time(), 0.0, {}
    folds = {'folded': 0, 'tripped': 0, 'leaves': 0}
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
        for prepared in (False, True):
            if prepared:
                c.um.plan(ctx).prepare(c.L, c.v, c.dt)
            got = mrtm.route_series(c.um, c.L, c.v, c.w.area, c.q, c.ndays, c.spin, S0=c.S0, dt=c.dt, flags=_hip.XH_ROUTE_REASSOC)
            if not prepared:
                kern = c.um.plan(ctx).info()['last_tree_kernel']
                kernels[kern] = kernels.get(kern, 0) + 1
            else:
                ri = c.um.plan(ctx).rsum_info()
                folds['folded'] += int(ri['folded'] > 0)
                folds['tripped'] += int(ri['fold_disabled'] > 0)
                folds['leaves'] += int(ri['folded'])
                tag = 'folded {:4d}{}'.format(int(ri['prepared_folded']), ' (guard tripped)' if ri['fold_disabled'] else '')
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
          'lane lags fall to the bit-exact lock-step kernel); prepared plans: {} routed with folded leaves ({} leaves), {} gave them '
          'up on a guard trip'.format(n, time.time() - t0, worst, kernels, folds['folded'], folds['leaves'], folds['tripped']))


if __name__ == '__main__':
    main()
