"""Per-unit cycle accounting of the dataflow routing kernel (run with XH_FLOW_STATS=1 on the GPU box)."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
os.environ['XH_FLOW_STATS'] = '1'
from xanthos_amd import _hip, synth
from xanthos_amd.pipeline import pipeline_from_world

months = int(sys.argv[1]) if len(sys.argv) > 1 else 120
ctx = _hip.get_context(0)
ncell_env = int(os.environ.get('XH_STATS_NCELL', '67420'))
w = synth.make_world(ncell=ncell_env, n_basins=max(1, 235 * ncell_env // 67420))
um = None
if os.environ.get('XH_STATS_SHARD'):      # "rank/ranks": one shard of the basin / network-closed partition (xanthos_amd.dist)
    from xanthos_amd.dist import make_shards, sub_world
    from xanthos_amd.pipeline import topology_from_world
    rk, nr = (int(x) for x in os.environ['XH_STATS_SHARD'].split('/'))
    um0 = topology_from_world(w)
    w, um = sub_world(w, um0, make_shards(w, um0, nr)[rk])
    print('shard', rk, 'of', nr, ':', w.ncell, 'cells')
pipe = pipeline_from_world(ctx, w, months, 1961, int(os.environ.get('XH_STATS_ABCD_SPIN', '60')), int(os.environ.get('XH_STATS_ROUTE_SPIN', '0')), um=um)
f = pipe.alloc_forcing()
ctx.synth_forcing(1, pipe.ncell, pipe.nmonths, ctx.upload(w.latitude), f, nan_frac=0.0)
pipe.run(('pm', 'abcd'))
for rep in range(2):
    if os.environ.get('XH_STATS_WITH_PM'):
        pipe.run(('pm', 'abcd'))
    ctx.timing_reset()
    pipe.run_mrtm()
    ms, n = ctx.timing('mrtm_route')
if os.environ.get('XH_STATS_LOOP'):
    ctx.sync(); ctx.timing_reset()
    for rep in range(int(os.environ['XH_STATS_LOOP'])):
        pipe.run(tuple(os.environ.get('XH_STATS_STAGES', 'pm,abcd,mrtm').split(',')))
        if os.environ.get('XH_STATS_SYNC'):
            ctx.sync()
    ctx.sync()
    ms, n = ctx.timing('mrtm_route')
    print('back-to-back steps: mrtm_route avg ms', ms / n)
st = pipe.plan.stats().astype(np.float64)
nsub = sum(int(d) * 8 for d in pipe.ndays)
print('plan', pipe.plan.info(), 'reassociated', pipe.plan.rsum_info())
print('route ms', ms / n, 'substeps', nsub, 'us/substep', ms / n * 1e3 / nsub)
raw3 = pipe.plan.stats()[:, 3]
if os.environ.get('XH_STATS_SAVE'):
    np.save(os.environ['XH_STATS_SAVE'], pipe.plan.stats())
loop, total, ticks, shape = st[:, 0], st[:, 1], st[:, 2], (raw3 & np.uint64(255)).astype(int)
clock = total / (ticks / 100e6) / 1e9
print('units', len(st), 'clock GHz median', np.median(clock))
print('loop cycles/substep: median %.0f  p10 %.0f  p90 %.0f  max %.0f' % tuple(np.percentile(loop / nsub, [50, 10, 90, 100])))
print('loop share of unit time: median %.2f min %.2f' % (np.median(loop / total), (loop / total).min()))
print('unit wall ms: median %.2f max %.2f' % (np.median(ticks / 1e5), ticks.max() / 1e5))
for wu in range(3, 10):
    for flag, name in ((0, 'isolated'), (16, 'imports'), (32, 'exports'), (48, 'both')):
        sel = ((shape & 15) == wu) & ((shape & 48) == flag)
        if sel.any():
            print('%d gathered terms %-9s n=%4d loop cyc/substep median %.0f' % (wu - 1, name, sel.sum(), np.median(loop[sel] / nsub)))
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

# placement: HW_ID bits (gfx9): wave_id[3:0] simd_id[5:4] pipe[7:6] cu_id[11:8] sh_id[12] se_id[15:13] ... ; XCC id at bit 40
hw = (raw3 >> np.uint64(8)) & np.uint64(0xffffffff)
xcc = ((raw3 >> np.uint64(40)) & np.uint64(15)).astype(int)
simd = ((hw >> np.uint64(4)) & np.uint64(3)).astype(int)
cu = ((hw >> np.uint64(8)) & np.uint64(15)).astype(int)
sh = ((hw >> np.uint64(12)) & np.uint64(1)).astype(int)
se = ((hw >> np.uint64(13)) & np.uint64(7)).astype(int)
cu_key = ((xcc * 8 + se) * 2 + sh) * 16 + cu
simd_key = cu_key * 4 + simd
per_cu = np.bincount(cu_key)
per_simd = np.bincount(simd_key)
print('distinct CUs used', (per_cu > 0).sum(), 'units per CU histogram', np.bincount(per_cu[per_cu > 0]))
print('waves per SIMD histogram', np.bincount(per_simd[per_simd > 0]))
share = per_simd[simd_key]
cps = loop / nsub
for k in sorted(set(share)):
    sel = share == k
    print('units on a SIMD holding %d wave(s): n=%d loop cyc/substep median %.0f max %.0f' % (k, sel.sum(), np.median(cps[sel]), cps[sel].max()))
idx = np.argsort(-cps)[:10]
print('slowest units: index, cyc/substep, shape(terms+1 | 16 imports | 32 outlets), waves on its SIMD, units on its CU')
for i in idx:
    print(' ', i, round(cps[i]), shape[i], share[i], per_cu[cu_key[i]])
# which units share a SIMD (placement of the blocks past the first 1024)
order = np.argsort(simd_key, kind='stable')
pairs = [(order[i], order[i + 1]) for i in range(len(order) - 1) if simd_key[order[i]] == simd_key[order[i + 1]]]
print('units sharing a SIMD (first 24 pairs):', pairs[:24])
lo = np.array([min(p) for p in pairs]); hi = np.array([max(p) for p in pairs])
if len(pairs):
    print('partner index: low min/max', lo.min(), lo.max(), ' high min/max', hi.min(), hi.max())
wall = ticks / 1e5
idx = np.argsort(-wall)[:10]
print('longest unit walls (ms):', [(int(i), round(float(wall[i]), 2), int(shape[i]), int(share[i])) for i in idx])
print('xcc of first 16 units', xcc[:16], 'cu', cu[:16], 'simd', simd[:16], 'se', se[:16])
terms = shape & 15
for k in sorted(set(per_cu[cu_key])):
    for t in sorted(set(terms)):
        sel = (per_cu[cu_key] == k) & (terms == t) & (share == 1)
        if sel.sum() >= 3:
            print('units on a CU holding %d, alone on their SIMD, %d gathered terms: n=%3d loop cyc/substep median %.0f max %.0f' % (
                k, t - 1, sel.sum(), np.median(cps[sel]), cps[sel].max()))
plain = ((raw3 >> np.uint64(6)) & np.uint64(1)).astype(int)
for pl in (0, 1):
    for t in sorted(set(terms)):
        for flag, name in ((0, 'isolated'), (16, 'imports'), (32, 'exports'), (48, 'both')):
            sel = (plain == pl) & (terms == t) & (share == 1) & ((shape & 48) == flag)
            if sel.sum() >= 1:
                print('%s units, alone on their SIMD, %d values read, %-8s: n=%3d own cyc/substep median %.0f min %.0f max %.0f' % (
                    'plain' if pl else 'pair ', t - 1, name, sel.sum(), np.median(cps[sel]), cps[sel].min(), cps[sel].max()))
# block index -> CU: which blocks share a CU
cu_blocks = {}
for i, ck in enumerate(cu_key):
    cu_blocks.setdefault(int(ck), []).append(i)
five = [v for v in cu_blocks.values() if len(v) == 5][:6]
four = [v for v in cu_blocks.values() if len(v) == 4][:6]
print('blocks of some 5-unit CUs', five)
print('blocks of some 4-unit CUs', four)
