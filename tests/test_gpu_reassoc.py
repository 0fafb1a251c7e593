"""GPU tests (-m gpu) of the REASSOCIATED ("tolerance") form of the routing kernel: XH_ROUTE_REASSOC, k_mrtm_rsum.

The form keeps the VALUE of every row sum of mrtm.py:50-51 but not its order (running sums along chains of lanes) and fuses
the update of mrtm.py:54-69, so its results equal the reference's to rounding, not bit for bit.  The bar written here:
identical NaN masks and |x - ref| <= 1e-9 |ref| + atol on every routed value (atol 1e-3 m3 for storages, 1e-9 m3/s for
flows: far below anything a cell holds or passes) against the golden vectors of the real reference and against the oracle.
The north star's gate is 1e-6.  This form is the library's default since round 5; the bit-exact kernels (XH_ROUTE_EXACT) stay
the checker (XH_ROUTE_VALIDATE).  Round 6: prepared plans pass ONE running sum per lane (the tests at the end of this file).
"""
from types import SimpleNamespace as NS

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

REASSOC, EXACT, VALIDATE = 128, 256, 32


def routed_close(x, ref, atol, rtol=1e-9, tag=''):
    x, ref = np.asarray(x), np.asarray(ref)
    assert x.shape == ref.shape
    assert np.array_equal(np.isnan(x), np.isnan(ref)), 'NaN pattern differs ' + str(tag)
    m = ~np.isnan(ref)
    excess = np.abs(x[m] - ref[m]) - (atol + rtol * np.abs(ref[m]))
    assert (excess <= 0).all(), '{}: {} values beyond the bar, largest excess {:.3e}'.format(tag, int((excess > 0).sum()), excess.max())
    return float(np.max(np.abs(x[m] - ref[m]) / np.maximum(np.abs(ref[m]), atol * 1e6))) if m.any() else 0.0


def series_close(got, ref, tag=''):
    chs, avg, fend = got
    routed_close(chs, ref[0], 1e-3, tag=(tag, 'chs'))
    routed_close(avg, ref[1], 1e-9, tag=(tag, 'avg'))
    routed_close(fend, ref[2], 1e-9, tag=(tag, 'F_end'))


@pytest.fixture(scope='module')
def hip():
    from xanthos_amd import _hip
    assert _hip.device_count() > 0, 'no GPU visible'
    assert _hip.XH_ROUTE_REASSOC == REASSOC and _hip.XH_ROUTE_EXACT == EXACT
    return _hip


def _um(t, tag):
    from xanthos_amd.routing import mrtm
    return mrtm.UpstreamMatrix(t[tag + '_um_indptr'], t[tag + '_um_indices'], t[tag + '_um_data'])


@pytest.mark.parametrize('tag', ['rand', 'tree'])
def test_route_series_golden_reassoc(hip, golden, tag):
    """The reference's own Components.calculate_routing (tests/golden/mrtm.npz): within the bar.  'rand' is not a forest, so the
    flag leaves it to the bit-exact kernels -- which is the contract: the flag never changes WHICH cells are routed."""
    from xanthos_amd.routing import mrtm
    g, t = golden('mrtm'), golden('topo')
    um = _um(t, tag)
    got = mrtm.route_series(um, g[tag + '_L'], g[tag + '_chv'], g[tag + '_area'], g[tag + '_series_runoff'],
                            g[tag + '_series_ndays'], int(g['series_spinup']), flags=REASSOC)
    series_close(got, (g[tag + '_series_chstorage'], g[tag + '_series_avgchflow'], g[tag + '_series_Fend']), tag)
    info = um.plan(hip.get_context()).info()
    if info['flow_cells'] > 0 and tag == 'tree':
        assert info['last_tree_kernel'] == 4, info


@pytest.mark.parametrize('tag', ['rand', 'tree'])
def test_streamrouting_golden_reassoc(hip, golden, tag):
    """One month at a time (the reference's streamrouting signature), 28 .. 31 days, state carried from call to call."""
    from xanthos_amd.routing import mrtm
    g, t = golden('mrtm'), golden('topo')
    um = _um(t, tag)
    S = g[tag + '_S0']
    n = len(S)
    for nday in (28, 29, 30, 31):
        S, favg, F = mrtm.streamrouting(g[tag + '_L'], S, np.zeros(n), g[tag + '_chv'], g['%s_q_%d' % (tag, nday)],
                                        g[tag + '_area'], nday, 10800, um, flags=REASSOC)
        routed_close(S, g['%s_S_%d' % (tag, nday)], 1e-3, tag=(tag, nday, 'S'))
        routed_close(favg, g['%s_Favg_%d' % (tag, nday)], 1e-9, tag=(tag, nday, 'Favg'))
        routed_close(F, g['%s_F_%d' % (tag, nday)], 1e-9, tag=(tag, nday, 'F'))
        S = g['%s_S_%d' % (tag, nday)]          # the next month starts from the reference's state, like the golden run


def _world(seed=3, ncell=3000, outlet_frac=0.02):
    from xanthos_amd import synth
    from xanthos_amd.routing import mrtm
    w = synth.make_world(nrow=60, ncol=120, ncell=ncell, n_basins=5, seed=seed, outlet_frac=outlet_frac)
    st = NS(ngridrow=w.nrow, ngridcol=w.ncol)
    um = mrtm.upstream_genmatrix(mrtm.upstream(w.coords, mrtm.downstream(w.coords, w.flow_dir, st), st))
    return w, um


def test_route_synthetic_world_reassoc_vs_oracle(hip):
    """3000 cells, networks far larger than one unit, NaN runoff cells, 2 + 12 months: k_mrtm_rsum against the oracle, against
    the bit-exact kernel of the same library, and validated on the device (XH_ROUTE_VALIDATE compares within 1e-9 then)."""
    from oracle import months as o_months
    from oracle import mrtm as o_mrtm
    from xanthos_amd.routing import mrtm
    w, um = _world()
    rng = np.random.default_rng(9)
    runoff = rng.gamma(2.0, 30.0, (w.ncell, 12))
    runoff[rng.random(w.ncell) < 0.01] = np.nan
    ndays = o_months.set_month_arrays(12, 1972, 1972)[:, 2]
    ref = o_mrtm.route_series(um.tocsr(), w.flow_dist, w.velocity, w.area, runoff, ndays, 2)
    got = mrtm.route_series(um, w.flow_dist, w.velocity, w.area, runoff, ndays, 2, flags=REASSOC)
    series_close(got, ref, 'oracle')
    plan = um.plan(hip.get_context())
    info = plan.info()
    assert info['last_tree_kernel'] == 4 and info['flow_cells'] == w.ncell and info['flow_edges'] > 10, info
    assert info['skew_max_lag'] <= 192
    exact = mrtm.route_series(um, w.flow_dist, w.velocity, w.area, runoff, ndays, 2, flags=EXACT)
    assert plan.info()['last_tree_kernel'] == 2
    for a, b in zip(exact, ref):
        assert np.array_equal(a, b, equal_nan=True)
    # the form does change bits (otherwise this test would not be looking at it) ...
    assert not np.array_equal(got[1], exact[1], equal_nan=True)
    # ... and the device-side cross-check accepts it within its tolerance
    n_val = plan.info()['validated']
    again = mrtm.route_series(um, w.flow_dist, w.velocity, w.area, runoff, ndays, 2, flags=REASSOC | VALIDATE)
    assert plan.info()['validated'] == n_val + 1 and plan.info()['last_tree_kernel'] == 4
    for a, b in zip(again, got):
        assert np.array_equal(a, b, equal_nan=True)          # the same kernel on the same input: the same bits


def test_route_reassoc_deep_chain_fires_and_initial_storage(hip):
    """A 1,500-cell main stem with side branches, scrambled ids, cells that fire every other sub-step, initial storage, 11
    months, spin-up 3: deep lane lags, every month boundary crossed lane by lane, partial output groups."""
    from oracle import mrtm as o_mrtm
    from xanthos_amd.routing import mrtm
    rng = np.random.default_rng(77)
    n_main, n = 1500, 1500 + 300
    ds = np.full(n, -1)
    ds[1:n_main] = np.arange(0, n_main - 1)
    ds[n_main:] = rng.integers(5, n_main, n - n_main)
    perm = rng.permutation(n)
    ds_p = np.full(n, -1)
    ds_p[perm] = np.where(ds >= 0, perm[np.clip(ds, 0, None)], -1)
    rows = [[] for _ in range(n)]
    for c in range(n):
        rows[c].append((c, -1))
        if ds_p[c] >= 0:
            rows[ds_p[c]].append((c, 1))
    indptr, indices, data = [0], [], []
    for r in rows:
        for col, sgn in sorted(r):
            indices.append(col)
            data.append(sgn)
        indptr.append(len(indices))
    um = mrtm.UpstreamMatrix(indptr, indices, data)
    L = rng.uniform(20e3, 60e3, n)
    L[rng.random(n) < 0.03] = 4e3
    v = rng.uniform(0.4, 1.5, n)
    area = rng.uniform(800, 3100, n)
    q = rng.gamma(2.0, 30.0, (n, 11))
    S0 = rng.uniform(0.0, 5e7, n)
    ndays = np.array([31, 28, 31, 30, 31, 30, 31, 31, 30, 31, 30])
    ref = o_mrtm.route_series(um.tocsr(), L, v, area, q, ndays, 3, S0=S0)
    got = mrtm.route_series(um, L, v, area, q, ndays, 3, S0=S0, flags=REASSOC)
    series_close(got, ref, 'deep chain')
    info = um.plan(hip.get_context()).info()
    assert info['last_tree_kernel'] == 4, info


@pytest.mark.parametrize('dt', [7200, 17280, 86400])
def test_route_reassoc_other_time_steps(hip, dt, tmp_path, monkeypatch):
    """dt = 2 h, 4.8 h (odd sub-step counts per month) and one day (months shorter than the lane lags: the flag then leaves the
    call to the lock-step kernel, bit-exact -- still within the bar; and the call is NOT on record as one routed by the
    reassociated kernel: ADVICE round 5 -- no first-check marker of that form is written for a kernel that never ran)."""
    from oracle import mrtm as o_mrtm
    from xanthos_amd.routing import mrtm
    monkeypatch.setenv('XH_CACHE_DIR', str(tmp_path / 'cache'))
    w, um = _world(seed=5)
    rng = np.random.default_rng(dt)
    q = rng.gamma(2.0, 30.0, (w.ncell, 5))
    ndays = np.array([31, 28, 31, 30, 31])
    ref = o_mrtm.route_series(um.tocsr(), w.flow_dist, w.velocity, w.area, q, ndays, 1, dt=dt)
    got = mrtm.route_series(um, w.flow_dist, w.velocity, w.area, q, ndays, 1, dt=dt, flags=REASSOC)
    series_close(got, ref, dt)
    assert um.plan(hip.get_context()).info()['last_tree_kernel'] == (4 if dt < 86400 else 1)
    marks = [f.name for f in (tmp_path / 'cache').iterdir() if f.name.startswith('route_ok_')] if (tmp_path / 'cache').exists() else []
    reassoc_marks = [m for m in marks if m.endswith(('_r', '_rf', '_rs'))]
    assert (len(reassoc_marks) == 1) == (dt < 86400), marks
    if dt == 86400:
        assert um.plan(hip.get_context()).rsum_info()['units'] == 0


def test_route_reassoc_fuzz(hip):
    """tools/fuzz_routing.py's generator, 25 seeded cases (12 .. 6,000 cells, month counts, spin-ups, time steps, NaN runoff,
    stagnant and over-fast channels, initial storage): the reassociated form against the oracle, within the bar."""
    import importlib.util
    import os
    from oracle import mrtm as o_mrtm
    from xanthos_amd.routing import mrtm
    spec = importlib.util.spec_from_file_location('fuzz_routing', os.path.join(os.path.dirname(__file__), '..', 'tools', 'fuzz_routing.py'))
    fz = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(fz)
    rng = np.random.default_rng(424242)
    used = []
    for k in range(25):
        c = fz.gen_case(rng)
        ref = o_mrtm.route_series(c.um.tocsr(), c.L, c.v, c.w.area, c.q, c.ndays, c.spin, S0=c.S0, dt=c.dt)
        got = mrtm.route_series(c.um, c.L, c.v, c.w.area, c.q, c.ndays, c.spin, S0=c.S0, dt=c.dt, flags=REASSOC)
        series_close(got, ref, (k, c.ncell, c.nm, c.spin, c.dt))
        used.append(c.um.plan(hip.get_context()).info()['last_tree_kernel'])
    assert used.count(4) >= 10, used


def test_fed_pipeline_with_reassoc_routing(hip):
    """The fed stage order (xh_run_fused mode 1) with the reassociated routing kernel: the same bits as the same kernel
    behind a stage-by-stage run, and all within the bar of the bit-exact routing of the same runoff."""
    from xanthos_amd import synth
    from xanthos_amd.pipeline import pipeline_from_world
    ctx = hip.get_context()
    w = synth.make_world(nrow=90, ncol=180, ncell=6000, n_basins=20, seed=11)
    nm = 240
    pipe = pipeline_from_world(ctx, w, nm, 1961, 25, 24, route_flags=REASSOC)
    ctx.synth_forcing(5, w.ncell, nm, ctx.upload(w.latitude), pipe.alloc_forcing(), nan_frac=0.002)
    pipe.run(fed=False)
    staged = pipe.download(('q', 'chs', 'avg'))
    assert pipe.plan.info()['last_tree_kernel'] == 4
    for k in ('chs', 'avg'):
        pipe.out[k].zero()
    pipe.run(fed=True)
    fed = pipe.download(('q', 'chs', 'avg'))
    assert pipe.plan.info()['last_tree_kernel'] == 4
    for k in ('q', 'chs', 'avg'):
        assert np.array_equal(staged[k], fed[k], equal_nan=True), k
    pipe.route_flags = EXACT
    pipe.run(fed=False)
    exact = pipe.download(('chs', 'avg'))
    assert pipe.plan.info()['last_tree_kernel'] == 2
    routed_close(fed['chs'], exact['chs'], 1e-3, tag='chs')
    routed_close(fed['avg'], exact['avg'], 1e-9, tag='avg')


def test_run_model_default_routing_form_against_the_oracle_chain(tmp_path):
    """run_model() as a user meets it since round 5 -- no flag, no environment switch, so the reassociated routing form (in a
    child process, so that nothing this test process has set is in the way) -- against the oracle chain PM -> ABCD ->
    MRTM: PET / AET / Q / Sav as before, ChStorage / Avg_ChFlow within the form's bar; and ``routing_form = exact`` in the ini
    gives the bit-exact kernels back."""
    import json
    import os
    import subprocess
    import sys
    from oracle import abcd as o_abcd, months as o_months, mrtm as o_mrtm, pm as o_pm
    from xanthos_amd import synth
    root = os.path.abspath(os.path.join(os.path.dirname(__file__), '..'))
    w = synth.make_world(nrow=36, ncol=72, ncell=900, n_basins=7, seed=33)
    f = synth.make_forcing(w, 36)
    child = r"""
import json, sys
sys.path.insert(0, sys.argv[1])
import numpy as np
from xanthos_amd import Xanthos
res = Xanthos(sys.argv[2]).execute()
np.savez(sys.argv[3], q=res.Q, chs=res.ChStorage, avg=res.Avg_ChFlow, area=res.data.area, L=res.data.flow_dist, v=res.data.str_velocity)
print(json.dumps({'kernel': int(res.pipe.plan.info()['last_tree_kernel'])}))
"""
    script = tmp_path / 'child.py'
    script.write_text(child)
    got = {}
    for form in ('default', 'exact'):
        d = str(tmp_path / form)
        os.makedirs(d)
        ini = synth.write_example(d, w, f, 1971, 1973, runoff_spinup=25, routing_spinup=6)
        if form == 'exact':
            text = open(ini).read()
            open(ini, 'w').write(text.replace('routing_spinup', 'routing_form = exact\nrouting_spinup', 1))
        env = dict(os.environ)
        env.pop('XH_ROUTE_REASSOC', None)
        r = subprocess.run([sys.executable, str(script), root, ini, os.path.join(d, 'out.npz')], env=env, capture_output=True,
                           text=True, timeout=600)
        assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
        kernel = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith('{')][-1])['kernel']
        assert kernel == (4 if form == 'default' else 2), (form, kernel)
        got[form] = np.load(os.path.join(d, 'out.npz'))
    st = NS(ngridrow=w.nrow, ngridcol=w.ncol)
    um = o_mrtm.upstream_genmatrix(o_mrtm.upstream(w.coords, o_mrtm.downstream(w.coords, w.flow_dir, st), st))
    ndays = o_months.set_month_arrays(36, 1971, 1973)[:, 2]
    g = got['exact']
    chs, avg, _ = o_mrtm.route_series(um, g['L'], g['v'], g['area'], g['q'], ndays, 6)
    assert np.array_equal(g['chs'], chs, equal_nan=True) and np.array_equal(g['avg'], avg, equal_nan=True)
    d_ = got['default']
    assert np.array_equal(d_['q'], g['q'], equal_nan=True)                       # the stages in front are the same kernels
    routed_close(d_['chs'], chs, 1e-3, tag='ChStorage')
    routed_close(d_['avg'], avg, 1e-9, tag='Avg_ChFlow')
    assert not np.array_equal(d_['avg'], avg, equal_nan=True)


def test_route_reassoc_edge_sizes_and_forced_fault(hip):
    """Smallest legal problems in the reassociated form -- a two-cell network (cell 1 drains into cell 0, a reach shorter than
    velocity x dt: it fires), 13 months (a partial group of output months), with and without spin-up; a single cell -- and the
    fault path: XH_ROUTE_TEST_FAULT raises the fault word in front of the launch, the units that wait give up, the call is
    routed again by the workgroup-per-network kernel at the next synchronisation (bit-exact then, hence inside the bar)."""
    from oracle import mrtm as o_mrtm
    from xanthos_amd.routing import mrtm
    um = mrtm.UpstreamMatrix([0, 2, 3], [0, 1, 1], [-1, 1, -1])
    L, v, area = np.array([30e3, 1000.0]), np.array([1.0, 2.0]), np.array([2500.0, 2400.0])
    q = np.random.default_rng(2).gamma(2.0, 30.0, (2, 13))
    ndays = np.array([31, 28, 31, 30, 31, 30, 31, 31, 30, 31, 30, 31, 31])
    for spin in (0, 5):
        ref = o_mrtm.route_series(um.tocsr(), L, v, area, q, ndays, spin)
        series_close(mrtm.route_series(um, L, v, area, q, ndays, spin, flags=REASSOC), ref, ('two cells', spin))
        assert um.plan(hip.get_context()).info()['last_tree_kernel'] == 4
    um1 = mrtm.UpstreamMatrix([0, 1], [0], [-1])
    ref = o_mrtm.route_series(um1.tocsr(), L[:1], v[:1], area[:1], q[:1], ndays, 2)
    series_close(mrtm.route_series(um1, L[:1], v[:1], area[:1], q[:1], ndays, 2, flags=REASSOC), ref, 'one cell')
    w, umw = _world(seed=8)
    rng = np.random.default_rng(4)
    runoff = rng.gamma(2.0, 30.0, (w.ncell, 8))
    nd = ndays[:8]
    ref = o_mrtm.route_series(umw.tocsr(), w.flow_dist, w.velocity, w.area, runoff, nd, 2)
    plan = umw.plan(hip.get_context())
    r0 = plan.info()['reroutes']
    got = mrtm.route_series(umw, w.flow_dist, w.velocity, w.area, runoff, nd, 2, flags=REASSOC | hip.XH_ROUTE_TEST_FAULT)
    series_close(got, ref, 'forced fault')
    assert plan.info()['reroutes'] == r0 + 1
    got = mrtm.route_series(umw, w.flow_dist, w.velocity, w.area, runoff, nd, 2, flags=REASSOC)      # (backs off: no dataflow kernel)
    series_close(got, ref, 'after the fault')


_REASSOC_PARTITION_CHILD = r"""
import json, os, sys
import numpy as np
sys.path.insert(0, sys.argv[1])
from types import SimpleNamespace as NS
from oracle import months as o_months
from oracle import mrtm as o_mrtm
from xanthos_amd import _hip, synth
from xanthos_amd.routing import mrtm
w = synth.make_world(nrow=60, ncol=120, ncell=3000, n_basins=5, seed=3, outlet_frac=0.02)
st = NS(ngridrow=w.nrow, ngridcol=w.ncol)
um = mrtm.upstream_genmatrix(mrtm.upstream(w.coords, mrtm.downstream(w.coords, w.flow_dir, st), st))
rng = np.random.default_rng(11)
runoff = rng.gamma(2.0, 30.0, (w.ncell, 12))
ndays = o_months.set_month_arrays(12, 1973, 1973)[:, 2]
ref = o_mrtm.route_series(um.tocsr(), w.flow_dist, w.velocity, w.area, runoff, ndays, 2)
got = mrtm.route_series(um, w.flow_dist, w.velocity, w.area, runoff, ndays, 2, flags=_hip.XH_ROUTE_REASSOC)
worst = 0.0
for x, r, atol in zip(got, ref, (1e-3, 1e-9, 1e-9)):
    err = np.abs(x - r)
    assert (err <= 1e-9 * np.abs(r) + atol).all()
    worst = max(worst, float((err / np.maximum(np.abs(r), 1e6 * atol)).max()))
info = um.plan(_hip.get_context()).info()
print(json.dumps({'kernel': int(info['last_tree_kernel']), 'units': int(info['flow_units']), 'edges': int(info['flow_edges']),
                  'lag': int(info['skew_max_lag']), 'worst': worst}))
"""


@pytest.mark.parametrize('env', [{}, {'XH_FLOW_PIECE_CAP': '64'}, {'XH_FLOW_PIECE_CAP': '20'}, {'XH_FLOW_PIECE_CAP': '5'},
                                 {'XH_FLOW_RS': '16384', 'XH_FLOW_SPARE': '3'}, {'XH_ROUTE_FENCED': '1'},
                                 {'XH_FLOW_CHECK': '1', 'XH_FLOW_PIECE_CAP': '12'}])
def test_route_reassoc_partition_variants(env, tmp_path):
    """The reassociated planner under other piece capacities (64: few streams, long chains; 5: a stream per handful of cells,
    chains of pieces everywhere), ring sizes (larger than the launch would pick: a ring below its formula -- 2,048 sub-steps
    here -- starves this plan's longest stream until a bounded wait gives up and the call is re-routed, round 6) and spare
    workgroups, the fully fenced publication, and the
    planner's invariant checker inside the library: every variant within the bar of the oracle (a child process each: the
    switches are read when the library builds its plan)."""
    import json
    import os
    import subprocess
    import sys
    script = tmp_path / 'child.py'
    script.write_text(_REASSOC_PARTITION_CHILD)
    root = os.path.abspath(os.path.join(os.path.dirname(__file__), '..'))
    e = dict(os.environ)
    e.update(env)
    out = subprocess.run([sys.executable, str(script), root], env=e, capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stdout[-1500:] + out.stderr[-1500:]
    info = json.loads(out.stdout.strip().splitlines()[-1])
    assert info['kernel'] == 4 and info['worst'] < 1e-10, info
    if env.get('XH_FLOW_PIECE_CAP') == '5':
        assert info['edges'] > 300, info


_FOLD_CHILD = r"""
import json, os, sys
import numpy as np
sys.path.insert(0, sys.argv[1])
from oracle import mrtm as o_mrtm
from xanthos_amd import _hip, synth
from xanthos_amd.pipeline import pipeline_from_world
ctx = _hip.get_context(0)
w = synth.make_world(nrow=60, ncol=120, ncell=4000, n_basins=9, seed=21, outlet_frac=0.12)      # many small river networks
nm = 36
pipe = pipeline_from_world(ctx, w, nm, 1971, 25, 12)           # (makes the plan and hands it velocity / flow distance / dt)
ctx.synth_forcing(7, w.ncell, nm, ctx.upload(w.latitude), pipe.alloc_forcing(), nan_frac=0.003)
pipe.run(('pm', 'abcd'), fed=False)
q = pipe.out['q'].download()
ref = o_mrtm.route_series(pipe.um.tocsr(), w.flow_dist, w.velocity, w.area, q, pipe.ndays, 12)

def close(tag):
    worst = 0.0
    for k, r, atol in (('chs', ref[0], 1e-3), ('avg', ref[1], 1e-9)):
        x = pipe.out[k].download()
        assert np.array_equal(np.isnan(x), np.isnan(r)), (tag, k)
        m = ~np.isnan(r)
        err = np.abs(x[m] - r[m])
        assert (err <= 1e-9 * np.abs(r[m]) + atol).all(), (tag, k, float(err.max()))
        worst = max(worst, float((err / np.maximum(np.abs(r[m]), 1e6 * atol)).max()))
    return worst
out = {}
for k in ('chs', 'avg'):
    pipe.out[k].zero()
pipe.run_mrtm()
out['worst'] = close('folded')
out['info'] = pipe.plan.rsum_info()
out['kernel'] = int(pipe.plan.info()['last_tree_kernel'])
pipe.run(fed=True)                                              # the fed order on the folded plan
out['worst_fed'] = close('fed')
out['info_fed'] = pipe.plan.rsum_info()
# the end state of the run (future mode's hand-over: S_end / F_end per cell) of folded leaves comes from their carriers' lanes
d_S, d_F = ctx.empty(w.ncell), ctx.empty(w.ncell)
ctx.route_series(pipe.plan, nm, 12, pipe.ndays, 10800.0, pipe.d_flow_dist, pipe.d_velocity, pipe.d_area, pipe.out['q'], None,
                 pipe.out['chs'], pipe.out['avg'], d_S, d_F, 0)
ctx.sync()
out['info_end'] = pipe.plan.rsum_info()
for name, x, r, atol in (('S_end', d_S.download(), ref[0][:, -1], 1e-3), ('F_end', d_F.download(), ref[2], 1e-9)):
    assert np.array_equal(np.isnan(x), np.isnan(r)), name
    m = ~np.isnan(r)
    assert (np.abs(x[m] - r[m]) <= 1e-9 * np.abs(r[m]) + atol).all(), (name, float(np.abs(x[m] - r[m]).max()))
# the guard: NEGATIVE runoff in the row of a folded leaf is outside the argument that lets its parent's lane carry it (it
# could fire) -- the unit gives up, the call is routed again on the plan without folded leaves, the result is still right
leaves = np.nonzero(np.diff(pipe.um.indptr) == 1)[0]
q2 = q.copy()
q2[leaves, 5] = -3.0
d_q2 = ctx.upload(q2)
ref = o_mrtm.route_series(pipe.um.tocsr(), w.flow_dist, w.velocity, w.area, q2, pipe.ndays, 12)
pipe.run_mrtm(runoff=d_q2)
ctx.sync()
out['worst_guard'] = close('guard')
out['info_guard'] = pipe.plan.rsum_info()
print(json.dumps(out))
"""


def test_folded_leaves_in_the_reassociated_form(tmp_path):
    """XH_FLOW_FOLD=1: leaves that cannot fire, in river networks small enough to have no streams, are carried by the lanes of
    their downstream cells (one fma recurrence each) -- fewer units, every value still within the bar of the oracle, staged
    and fed; and the guard: negative runoff in a folded leaf's row makes the unit give up, the call is routed again on the
    plan without folded leaves (which is then the one in use)."""
    import json
    import os
    import subprocess
    import sys
    script = tmp_path / 'fold_child.py'
    script.write_text(_FOLD_CHILD)
    root = os.path.abspath(os.path.join(os.path.dirname(__file__), '..'))
    env = dict(os.environ, XH_FLOW_FOLD='1', XH_FLOW_CHECK='1')
    env.pop('XH_ROUTE_REASSOC', None)
    r = subprocess.run([sys.executable, str(script), root], env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-3000:]
    out = json.loads(r.stdout.strip().splitlines()[-1])
    assert out['kernel'] == 4 and out['worst'] < 1e-10 and out['worst_fed'] < 1e-10 and out['worst_guard'] < 1e-10, out
    assert out['info']['folded'] > 50 and out['info_fed']['folded'] == out['info']['folded'] == out['info_end']['folded'], out
    assert out['info_guard']['fold_disabled'] == 1 and out['info_guard']['folded'] == 0, out


# ---------------------------------------------------------------------------------------------------------------------------
# Single running sums (round 6): prepared plans pass ONE sum per lane; the cells that may fire and have an upstream neighbour
# that may -- and a halo below them -- keep the pair form in units of their own (xh_flow_rsum.cpp; tests/corner_world.py).

_SINGLE_CHILD = r"""
import json, os, sys
import numpy as np
sys.path.insert(0, sys.argv[1])
sys.path.insert(0, os.path.join(sys.argv[1], 'tests'))
import corner_world as cw
from oracle import mrtm as o_mrtm
from xanthos_amd import _hip
from xanthos_amd.routing import mrtm
seed, mode = int(sys.argv[2]), sys.argv[3]
idx, csr, L, v, area = cw.make(seed=seed)
n = len(L)
q = cw.runoff(n, seed=seed)
if mode == 'negative_runoff':
    q[idx['s0_t1_0'], 3] = -5.0                     # outside the argument (lateral inflow >= 0): the guard must trip
ndays = np.array([31, 28, 31, 30, 31, 30])
um = mrtm.UpstreamMatrix(*csr)
fired, neg_s, neg_f = cw.instrumented(csr, L, v, area, q, ndays)
ref = o_mrtm.route_series(um.tocsr(), L, v, area, q, ndays, 0)
got = mrtm.route_series(um, L, v, area, q, ndays, 0)
plan = um.plan(_hip.get_context())
worst = 0.0
for x, r, atol in zip(got, ref, (1e-3, 1e-9, 1e-9)):
    assert np.array_equal(np.isnan(x), np.isnan(r))
    err = np.abs(x - r)
    assert (err <= 1e-9 * np.abs(r) + atol).all(), float((err - 1e-9 * np.abs(r)).max())
    worst = max(worst, float((err / np.maximum(np.abs(r), 1e6 * atol)).max()))
info = plan.info()
print(json.dumps({'kernel': int(info['last_tree_kernel']), 'rsum': plan.rsum_info(), 'worst': worst, 'reroutes': int(info['reroutes']),
                  'guard_trips': int(plan.rsum_info()['guard_trips']),
                  'neg_storage_cells': int((neg_s > 0).sum()), 'fired_unexpectedly': int(((fired > 0) & (v / L * 10800.0 < 1)).sum()),
                  'neg_s_sites': [int(neg_s[idx['s%d_A' % k]]) for k in range(3)]}))
"""


def _single_child(tmp_path, seed, mode, env=None):
    import json
    import os
    import subprocess
    import sys
    script = tmp_path / 'single_child.py'
    script.write_text(_SINGLE_CHILD)
    root = os.path.abspath(os.path.join(os.path.dirname(__file__), '..'))
    e = dict(os.environ, XH_FLOW_CHECK='1')
    e.pop('XH_ROUTE_REASSOC', None)
    e.update(env or {})
    r = subprocess.run([sys.executable, str(script), root, str(seed), mode], env=e, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-3000:]
    return json.loads(r.stdout.strip().splitlines()[-1])


@pytest.mark.parametrize('seed', [3, 13, 20])
def test_single_sum_plan_on_the_corner_world(tmp_path, seed):
    """The reference's corner S1 >= 0 > S2 (mrtm.py:54, :66-69: negative storage, negative outflow, cells downstream firing
    that cannot by construction) on the committed world, routed through the plugin API -- ``routing.mrtm.route_series`` prepares
    the plan from the L, ChV and dt it holds -- on the single-sum plan: within the bar of the oracle, no guard trip."""
    out = _single_child(tmp_path, seed, 'plain')
    assert out['neg_storage_cells'] >= 1 and out['fired_unexpectedly'] >= 2, out      # the corner does occur in this world
    assert out['kernel'] == 4 and out['worst'] < 1e-10, out
    assert out['rsum']['pair_cells'] > 0 and out['rsum']['fold_disabled'] == 0 and out['guard_trips'] == 0, out


def test_single_sum_guards_trip_and_the_call_is_rerouted(tmp_path):
    """Forced guard trips: (a) no halo (XH_RSUM_HALO=0): the negative outflow of the corner cell leaves the pair units, the exit
    guard trips; (b) negative runoff in a single unit.  Either way the call is routed again on the plan of pairs -- results
    within the bar -- and the prepared plan is switched off."""
    a = _single_child(tmp_path, 3, 'plain', {'XH_RSUM_HALO': '0'})
    assert a['kernel'] == 4 and a['worst'] < 1e-10, a
    assert a['rsum']['fold_disabled'] == 1 and a['rsum']['pair_cells'] == -1 and a['guard_trips'] >= 1, a
    b = _single_child(tmp_path, 3, 'negative_runoff')
    assert b['kernel'] == 4 and b['worst'] < 1e-10, b
    assert b['rsum']['fold_disabled'] == 1 and b['rsum']['pair_cells'] == -1 and b['guard_trips'] >= 1, b
