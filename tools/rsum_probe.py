"""Reassociated ("tolerance") routing form against the bit-exact kernel on one world (GPU box).

    python tools/rsum_probe.py [months] [routing spin-up] [reps]

Routes the same runoff with XH_ROUTE_EXACT and XH_ROUTE_REASSOC, compares every routed value (NaN masks, largest relative
difference, values beyond 1e-9), times `mrtm_route` for both forms alternately and prints the per-unit cycle accounting of
the reassociated launch (XH_FLOW_STATS=1).  XH_PROBE_NCELL: a smaller world.
"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
os.environ.setdefault('XH_FLOW_STATS', '1')
from xanthos_amd import _hip, synth                      # noqa: E402
from xanthos_amd.pipeline import pipeline_from_world     # noqa: E402

months = int(sys.argv[1]) if len(sys.argv) > 1 else 600
spin = int(sys.argv[2]) if len(sys.argv) > 2 else 120
reps = int(sys.argv[3]) if len(sys.argv) > 3 else 5
ncell = int(os.environ.get('XH_PROBE_NCELL', '67420'))
ctx = _hip.get_context(0)
w = synth.make_world(ncell=ncell, n_basins=max(1, 235 * ncell // 67420))
pipe = pipeline_from_world(ctx, w, months, 1961, min(120, months), spin)
f = pipe.alloc_forcing()
ctx.synth_forcing(1, pipe.ncell, pipe.nmonths, ctx.upload(w.latitude), f, nan_frac=float(os.environ.get('XH_PROBE_NAN', '0.001')))
pipe.run(('pm', 'abcd'), fed=False)
ctx.sync()


def route(flags):
    pipe.route_flags = flags
    ctx.timing_reset()
    pipe.run_mrtm()
    ctx.sync()
    ms, n = ctx.timing('mrtm_route')
    return ms / max(n, 1)


res = {}
for name, flags in (('exact', _hip.XH_ROUTE_EXACT), ('reassoc', _hip.XH_ROUTE_REASSOC)):
    ms = route(flags)
    info = pipe.plan.info()
    res[name] = {k: pipe.out[k].download().copy() for k in ('chs', 'avg')}
    print('%-8s first call %.2f ms; routed by kernel %d; units %d, streams %d, depth %d, max lag %d' % (
        name, ms, info['last_tree_kernel'], info['flow_units'], info['flow_edges'], info['flow_depth'], info['skew_max_lag']), flush=True)
print('reassociated plan:', pipe.plan.rsum_info(), flush=True)
ok = True
for k in ('chs', 'avg'):
    a, b = res['reassoc'][k], res['exact'][k]
    nan_same = bool((np.isnan(a) == np.isnan(b)).all())
    m = ~np.isnan(b)
    err = np.abs(a[m] - b[m])
    ref = np.abs(b[m])
    scale = ref.max()
    rel = err / np.maximum(ref, 1e-300)
    sig = ref > 1e-12 * scale
    far = int((err > 1e-9 * ref + 1e-12 * scale).sum())
    print('%s: NaN masks equal %s (%d NaN); max rel (|ref| > 1e-12 max) %.3e; max abs %.3e at scale %.3e; values beyond 1e-9: %d of %d; '
          'zeros in ref %d, of them nonzero here %d' % (k, nan_same, int((~m).sum()), rel[sig].max(), err.max(), scale, far, m.sum(),
                                                        int((ref == 0).sum()), int(((ref == 0) & (err > 0)).sum())))
    ok = ok and nan_same and far == 0
print('PARITY', 'ok' if ok else 'FAILED', flush=True)

t = {'exact': [], 'reassoc': []}
for r in range(reps):
    for name, flags in (('exact', _hip.XH_ROUTE_EXACT), ('reassoc', _hip.XH_ROUTE_REASSOC)):
        t[name].append(route(flags))
for name in t:
    print('%-8s mrtm_route ms: %s' % (name, ' '.join('%.2f' % x for x in t[name])))

# per-unit accounting of the last reassociated launch
route(_hip.XH_ROUTE_REASSOC)
st = pipe.plan.stats().astype(np.float64)
raw3 = pipe.plan.stats()[:, 3]
nsub = (sum(int(d) for d in pipe.ndays) + sum(int(d) for d in pipe.ndays[:spin])) * 8
loop, total, ticks = st[:, 0], st[:, 1], st[:, 2]
shape = (raw3 & np.uint64(255)).astype(int)
clock = total / (ticks / 100e6) / 1e9
print('units', len(st), 'sub-steps', nsub, 'clock GHz median %.3f' % np.median(clock))
print('cycles per sub-step outside waits: median %.0f p10 %.0f p90 %.0f max %.0f' % tuple(np.percentile(loop / nsub, [50, 10, 90, 100])))
print('cycles per sub-step of wall:       median %.0f max %.0f' % (np.median(total / nsub), (total / nsub).max()))
print('unit wall ms: median %.2f max %.2f' % (np.median(ticks / 1e5), ticks.max() / 1e5))
for flag, name in ((0, 'no streams'), (16, 'imports'), (32, 'exports'), (48, 'both')):
    sel = (shape & 48) == flag
    if sel.any():
        print('  %-10s n=%4d cycles/sub-step median %.0f max %.0f; wait data %.1f%% ring %.1f%%' % (
            name, sel.sum(), np.median(loop[sel] / nsub), (loop[sel] / nsub).max(), 100 * np.median(st[sel, 4] / total[sel]),
            100 * np.median(st[sel, 5] / total[sel])))
single_plan = pipe.plan.rsum_info()['pair_cells'] >= 0      # bit 64: 8-byte entries = a single unit; the others are its pair units
pairu = ((shape & 64) == 0) if single_plan else np.zeros(len(shape), dtype=bool)
if pairu.any():
    print('  pair units of the single-sum plan: n=%d cycles/sub-step outside waits %s, of wall %s' % (
        pairu.sum(), ' '.join('%.0f' % x for x in loop[pairu] / nsub), ' '.join('%.0f' % x for x in total[pairu] / nsub)))
    hwp = (raw3 >> np.uint64(8)) & np.uint64(0xffffffff)
    cu_key = ((raw3 >> np.uint64(40)) & np.uint64(15)).astype(np.int64) * 65536 + ((hwp >> np.uint64(8)) & np.uint64(0xff)).astype(np.int64)
    print('  units on the CUs of pair units (1 each = a CU of its own):', [int((cu_key == k).sum()) for k in cu_key[pairu]])
for reads in sorted(set(shape & 15)):
    sel = (shape & 15) == reads
    print('  LDS ops per sub-step %d: n=%4d cycles/sub-step median %.0f max %.0f' % (reads, sel.sum(), np.median(loop[sel] / nsub), (loop[sel] / nsub).max()))
hw = (raw3 >> np.uint64(8)) & np.uint64(0xffffffff)
key = ((raw3 >> np.uint64(40)) & np.uint64(15)).astype(np.int64) * 65536 + ((hw >> np.uint64(4)) & np.uint64(0xfff)).astype(np.int64)
uniq, cnt = np.unique(key, return_counts=True)
shared = np.isin(key, uniq[cnt > 1])
print('SIMDs in use %d, with two units %d; units on shared SIMDs: cycles/sub-step of wall median %.0f max %.0f' % (
    len(uniq), int((cnt > 1).sum()), np.median(total[shared] / nsub) if shared.any() else 0, (total[shared] / nsub).max() if shared.any() else 0))
zone = (raw3 >> np.uint64(44)).astype(int)
print('boundary groups per unit: median %d max %d of %d groups' % (np.median(zone), zone.max(), nsub // 16))
order = np.argsort(-loop)
print('slowest units (cycles/sub-step outside waits, wall, streams, boundary groups, shared SIMD):')
for u in order[:10]:
    print('  unit %4d  %.0f  %.0f  %s  %d  %s' % (u, loop[u] / nsub, total[u] / nsub, {0: '-', 16: 'in', 32: 'out', 48: 'in+out'}[shape[u] & 48],
                                               zone[u], 'shared' if shared[u] else ''))
sys.exit(0 if ok else 3)
