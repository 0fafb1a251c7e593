"""Config 5, objective only: 512-member population x 235 basins (set_calibrate = 0, km3_per_mth), as one multi-basin launch
and as 235 per-basin launches.  Superseded as the bench line by `bench.py --workload calib` (device-side DE generation on
real PM PET); kept for the per-basin vs multi-basin comparison quoted in DESIGN.md."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
from xanthos_amd import _hip, synth

members = int(sys.argv[1]) if len(sys.argv) > 1 else 512
nm, spin = 480, 120
ctx = _hip.get_context(0)
w = synth.make_world()
f = {k: ctx.empty((w.ncell, nm)) for k in synth.FORCING_NAMES}
ctx.synth_forcing(5, w.ncell, nm, ctx.upload(w.latitude), f, nan_frac=0.0)
rng = np.random.default_rng(0)
pars = np.stack([rng.uniform(1e-4, 1 - 1e-4, members), rng.uniform(1e-4, 8 - 1e-4, members),
                 rng.uniform(1e-4, 1 - 1e-4, members), rng.uniform(1e-4, 1 - 1e-4, members),
                 rng.uniform(1e-4, 1 - 1e-4, members)], axis=1)
order = np.argsort(w.basin_ids, kind='stable')
counts = np.bincount(w.basin_ids, minlength=w.n_basins + 1)[1:]
start = np.concatenate([[0], np.cumsum(counts)])
obs = rng.uniform(5, 60, nm)
# per basin: forcing rows gathered and transposed to [month, cell] once (reused by every generation);
# 'rsds' (30..330) stands in for PET: any positive field of that magnitude exercises the same arithmetic
blocks = []
for b in range(w.n_basins):
    n = int(counts[b])
    blk = {}
    src_rows = ctx.upload(order[start[b]:start[b + 1]], dtype=np.int64)
    for k in ('rsds', 'precip', 'abcd_tmin'):
        tmp = ctx.empty((n, nm)); ctx.gather_rows(f[k], src_rows, n, nm, tmp)
        t = ctx.empty((nm, n)); ctx.transpose(tmp, n, nm, t)
        blk[k] = t; tmp.free()
    src_rows.free()
    blk['area'] = ctx.upload(w.area[order[start[b]:start[b + 1]]])
    blocks.append(blk)
ctx.sync()
for rep in range(2):
    ctx.timing_reset()
    t0 = time.perf_counter()
    for b, blk in enumerate(blocks):
        ed = ctx.calib_objective(int(counts[b]), nm, spin, pars, blk['rsds'], blk['precip'], blk['abcd_tmin'], blk['area'], obs)
    dt = time.perf_counter() - t0
ms_a, n_a = ctx.timing('calib_abcd'); ms_k, n_k = ctx.timing('calib_kge')
# all basins in ONE launch (xh_calib_objective_multi): each basin with its own population
pars_nb = np.broadcast_to(pars, (w.n_basins,) + pars.shape).copy()
obs_nb = np.broadcast_to(obs, (w.n_basins, nm)).copy()
lst = lambda k: [blk[k] for blk in blocks]
for rep in range(2):
    ctx.timing_reset()
    t0 = time.perf_counter()
    ed_multi = ctx.calib_objective_multi(counts, nm, spin, pars_nb, lst('rsds'), lst('precip'), lst('abcd_tmin'), lst('area'), obs_nb)
    dt_multi = time.perf_counter() - t0
ms_m, _ = ctx.timing('calib_abcd')
assert np.array_equal(ed_multi[-1], ed), 'multi-basin launch differs from the per-basin launch'
print('same generation as ONE multi-basin launch: %.3f s wall (kernels %.1f ms) = %.3e member-cell-months/s incl. spin-up, %.0f objective evaluations/s' % (
    dt_multi, ms_m, members * w.ncell * (nm + spin) / dt_multi, members * w.n_basins / dt_multi))
mcm = members * w.ncell * nm
print('generation of %d members x %d basins: %.3f s wall, kernels abcd %.1f ms kge %.1f ms' % (members, w.n_basins, dt, ms_a, ms_k))
print('member-cell-months/s (sim months): %.3e ; incl. spin-up months: %.3e ; objective evaluations/s: %.0f' % (
    mcm / dt, members * w.ncell * (nm + spin) / dt, members * w.n_basins / dt))
