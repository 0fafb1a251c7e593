"""Where the cycles of the time-skewed routing kernel go, per unit (library built with `make PROFILE=1`):
ordinary groups of 16 sub-steps, groups with a month boundary, checks + month bookkeeping.  Run on the GPU box."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
os.environ['XH_FLOW_STATS'] = '1'
from xanthos_amd import _hip, synth
# make -C xanthos_amd/csrc prof (bit-exact kernel) / make exprsum EXPNAME=rprof EXPFLAGS=-DXH_WAVE_PROFILE (default kernel)
_hip.LIB_PATH = os.path.join(os.path.dirname(_hip.LIB_PATH), os.environ.get('XH_PROFILE_LIB', 'libxanthos_hip_prof.so'))
from xanthos_amd.pipeline import pipeline_from_world

months = int(sys.argv[1]) if len(sys.argv) > 1 else 240
ctx = _hip.get_context(0)
w = synth.make_world()
pipe = pipeline_from_world(ctx, w, months, 1961, 60, 0, route_flags=int(os.environ.get('XH_PROFILE_ROUTE_FLAGS', '0')))
ctx.synth_forcing(1, pipe.ncell, pipe.nmonths, ctx.upload(w.latitude), pipe.alloc_forcing(), nan_frac=0.0)
pipe.run(('pm', 'abcd'))
for rep in range(2):
    ctx.timing_reset()
    pipe.run_mrtm()
    ms, n = ctx.timing('mrtm_route')
st = pipe.plan.stats()
raw3 = st[:, 3]
shape = (raw3 & np.uint64(255)).astype(int)
zg = (raw3 >> np.uint64(44)).astype(np.float64)
st = st.astype(np.float64)
nsub = sum(int(d) * 8 for d in pipe.ndays)
groups = nsub / 16.0
print('route ms', ms / n, 'reassociated', pipe.plan.rsum_info())
plain, total, zone, fin = st[:, 0], st[:, 1], st[:, 4], st[:, 5]
print('units', len(st), 'boundary groups per unit: median %.0f of %.0f' % (np.median(zg), groups))
if pipe.plan.info()['last_tree_kernel'] == 4:      # k_mrtm_rsum: by streams and entry size (bit 64: 8-byte entries = single units)
    for flag, name in ((0, 'no streams'), (16, 'imports'), (32, 'exports'), (48, 'both')):
        for pl in (0, 64):
            sel = ((shape & 48) == flag) & ((shape & 64) == pl)
            if not sel.any():
                continue
            og = groups - zg[sel]
            print('%-10s %-6s n=%4d | cycles per sub-step: ordinary groups %.0f (p90 %.0f), boundary groups %.0f (p90 %.0f) | per sub-step of the run: '
                  'ordinary %.0f boundary %.0f checks+months %.0f total %.0f (max %.0f)' % (
                      name, 'single' if pl else 'pair', sel.sum(), np.median(plain[sel] / og / 16), np.percentile(plain[sel] / og / 16, 90),
                      np.median(zone[sel] / zg[sel] / 16), np.percentile(zone[sel] / zg[sel] / 16, 90),
                      np.median(plain[sel] / nsub), np.median(zone[sel] / nsub), np.median(fin[sel] / nsub), np.median(total[sel] / nsub),
                      (total[sel] / nsub).max()))
    pu = np.nonzero((shape & 64) == 0)[0] if pipe.plan.rsum_info()['pair_cells'] >= 0 else []
    for u in pu:
        og = groups - zg[u]
        print('pair unit %4d: ordinary groups %.0f, boundary groups %.0f (%d of %d groups) | of the run: ordinary %.0f boundary %.0f checks+months %.0f total %.0f' % (
            u, plain[u] / og / 16, zone[u] / zg[u] / 16, zg[u], groups, plain[u] / nsub, zone[u] / nsub, fin[u] / nsub, total[u] / nsub))
    sys.exit(0)
for flag, name in ((0, 'isolated'), (16, 'imports'), (32, 'exports'), (48, 'both')):
    for pl in (0, 64):
        for reads in range(1, 10):
            sel = ((shape & 15) == reads + 1) & ((shape & 48) == flag) & ((shape & 64) == pl)
            if sel.sum() < 3:
                continue
            og = groups - zg[sel]
            print('%-8s %-5s %d reads n=%4d | cycles per sub-step: ordinary groups %.0f, boundary groups %.0f | per sub-step of the run: '
                  'ordinary %.0f boundary %.0f checks+months %.0f total %.0f' % (
                      name, 'plain' if pl else 'pair', reads, sel.sum(), np.median(plain[sel] / og / 16), np.median(zone[sel] / zg[sel] / 16),
                      np.median(plain[sel] / nsub), np.median(zone[sel] / nsub), np.median(fin[sel] / nsub), np.median(total[sel] / nsub)))
