"""Per-unit cycle accounting of the dataflow routing kernel (run with XH_FLOW_STATS=1 on the GPU box)."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
os.environ['XH_FLOW_STATS'] = '1'
from xanthos_amd import _hip, synth
from xanthos_amd.pipeline import pipeline_from_world

months = int(sys.argv[1]) if len(sys.argv) > 1 else 120
ctx = _hip.get_context(0)
w = synth.make_world()
pipe = pipeline_from_world(ctx, w, months, 1961, 60, 0)
f = pipe.alloc_forcing()
ctx.synth_forcing(1, pipe.ncell, pipe.nmonths, ctx.upload(w.latitude), f, nan_frac=0.0)
pipe.run(('pm', 'abcd'))
for rep in range(2):
    ctx.timing_reset()
    pipe.run_mrtm()
    ms, n = ctx.timing('mrtm_route')
st = pipe.plan.stats().astype(np.float64)
nsub = sum(int(d) * 8 for d in pipe.ndays)
print('route ms', ms / n, 'substeps', nsub, 'us/substep', ms / n * 1e3 / nsub)
loop, total, ticks, shape = st[:, 0], st[:, 1], st[:, 2], st[:, 3].astype(int)
clock = total / (ticks / 100e6) / 1e9
print('units', len(st), 'clock GHz median', np.median(clock))
print('loop cycles/substep: median %.0f  p10 %.0f  p90 %.0f  max %.0f' % tuple(np.percentile(loop / nsub, [50, 10, 90, 100])))
print('loop share of unit time: median %.2f min %.2f' % (np.median(loop / total), (loop / total).min()))
print('unit wall ms: median %.2f max %.2f' % (np.median(ticks / 1e5), ticks.max() / 1e5))
for wu in (3, 5, 7, 9):
    for flag, name in ((0, 'isolated'), (16, 'imports'), (32, 'exports'), (48, 'both')):
        sel = ((shape & 15) == wu) & ((shape & 48) == flag)
        if sel.any():
            print('WU=%d %-9s n=%4d loop cyc/substep median %.0f' % (wu, name, sel.sum(), np.median(loop[sel] / nsub)))
nit = pipe.nmonths + pipe.routing_spinup
ovh = (total - loop) / nit / clock / 1e3     # us per month outside the sub-step loops
for flag, name in ((0, 'isolated'), (16, 'imports'), (32, 'exports'), (48, 'both')):
    sel = (shape & 48) == flag
    if sel.any():
        print('%-9s n=%4d per-month overhead us: median %.1f p90 %.1f max %.1f ; loop us/month median %.1f' % (
            name, sel.sum(), np.median(ovh[sel]), np.percentile(ovh[sel], 90), ovh[sel].max(),
            np.median(loop[sel] / nit / clock[sel] / 1e3)))
wd, wr = st[:, 4] / nit / clock / 1e3, st[:, 5] / nit / clock / 1e3
for flag, name in ((0, 'isolated'), (16, 'imports'), (32, 'exports'), (48, 'both')):
    sel = (shape & 48) == flag
    if sel.any():
        print('%-9s wait-for-data us/month median %.1f p90 %.1f | wait-for-ring median %.1f p90 %.1f | other %.1f' % (
            name, np.median(wd[sel]), np.percentile(wd[sel], 90), np.median(wr[sel]), np.percentile(wr[sel], 90),
            np.median(ovh[sel] - wd[sel] - wr[sel])))
