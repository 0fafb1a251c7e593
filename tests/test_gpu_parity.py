"""GPU parity tests (-m gpu): the HIP path, called through the C-ABI, against (1) the golden vectors generated from
the real reference and (2) the numpy oracle on seeded synthetic inputs.

Tolerances: the bit-exact routing kernels (asked for by flag: XH_ROUTE_EXACT -- the library's default since round 5 is the
reassociated form) are bit-exact (same operation order as numpy/scipy, -ffp-contract=off) for identical runoff; tests
that do not pass the flag run whatever the library ships and hold it to the form's bar (`routed_close`: identical NaN
masks, |x - ref| <= 1e-9 |ref| + 1e-3 m3 / 1e-9 m3/s).  PM and ABCD are fp64 within 1e-9 relative (only exp/log/sqrt/pow
implementations differ, a few ulp) -- far inside the 1e-6 the north star states.
"""
from types import SimpleNamespace

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

RTOL = 1e-9
EXACT = 256      # XH_ROUTE_EXACT: the bit-exact kernels for this call, whatever the library's default form is


def routed_close(x, ref, atol, rtol=1e-9, tag=''):
    """The bar of the default (reassociated) routing form: identical NaN masks, |x - ref| <= rtol |ref| + atol."""
    x, ref = np.asarray(x), np.asarray(ref)
    assert x.shape == ref.shape
    assert np.array_equal(np.isnan(x), np.isnan(ref)), 'NaN pattern differs ' + str(tag)
    m = ~np.isnan(ref)
    excess = np.abs(x[m] - ref[m]) - (atol + rtol * np.abs(ref[m]))
    assert (excess <= 0).all(), '{}: {} values beyond the bar, largest excess {:.3e}'.format(tag, int((excess > 0).sum()), excess.max())


ROUTED_ATOL = {'chs': 1e-3, 'avg': 1e-9}      # m3 / m3 s-1


def outputs_close(got, ref, keys, tag=''):
    """Pipeline outputs: the stages in front of the routing bit for bit, the routed ones within the default form's bar."""
    for k in keys:
        if k in ROUTED_ATOL:
            routed_close(got[k], ref[k], ROUTED_ATOL[k], tag=(tag, k))
        else:
            assert np.array_equal(got[k], ref[k], equal_nan=True), (tag, k)


def close(x, ref, rtol=RTOL, atol=1e-9):
    x, ref = np.asarray(x), np.asarray(ref)
    assert x.shape == ref.shape
    assert np.array_equal(np.isnan(x), np.isnan(ref)), 'NaN pattern differs'
    m = ~np.isnan(ref)
    err = np.abs(x[m] - ref[m]) - (atol + rtol * np.abs(ref[m]))
    assert (err <= 0).all(), 'max excess {:.3e}'.format(err.max())


@pytest.fixture(scope='module')
def hip():
    from xanthos_amd import _hip
    assert _hip.device_count() > 0, 'no GPU visible'
    return _hip


PM_FIELDS = ('cL', 'beta', 'rslimit', 'ae', 'be', 'Tminopen', 'Tminclose', 'VPDclose', 'VPDopen', 'RBLmin', 'RBLmax',
             'rc', 'emiss', 'alpha', 'lai', 'laimax', 'laimin', 'elev', 'tair_load', 'TMIN_load', 'rhs_load',
             'wind_load', 'rsds_load', 'rlds_load', 'tairprev_load', 'lct_load')


def test_pm_golden(hip, golden):
    from xanthos_amd.pet import penman_monteith as pm
    g = golden('pm')
    d = SimpleNamespace(**{k: g[k] for k in PM_FIELDS})
    lcy = [int(v) for v in g['lc_years']]
    n = d.tair_load.shape[0]
    pet = pm.run_pmpet(d, n, int(g['nlcs']), int(g['start_year']), int(g['end_year']), 0, 6, lcy)
    close(pet, g['pet'])
    pet = pm.run_pmpet(d, n, int(g['nlcs']), int(g['start_year']), int(g['end_year']), 2, 7, lcy)
    close(pet, g['pet_alt'])
    with pytest.raises(IndexError):
        pm.run_pmpet(d, n, 6, int(g['start_year']), int(g['end_year']), 0, 5, lcy)


def test_pm_synthetic_vs_oracle(hip):
    from oracle import pm as o_pm
    from xanthos_amd import synth
    from xanthos_amd.pet import penman_monteith as pm
    w = synth.make_world(nrow=60, ncol=120, ncell=3000, n_basins=12, seed=77)
    f = synth.make_forcing(w, 120)                       # 1961-1970: leap years + land-cover switch
    d = synth.data_bag(w, f)
    ref = o_pm.run_pmpet(d, w.ncell, w.nlcs, 1961, 1970, 0, 6, w.lc_years)
    got = pm.run_pmpet(d, w.ncell, w.nlcs, 1961, 1970, 0, 6, w.lc_years)
    close(got, ref)
    # tairprev = NULL path: the library derives it from the previous cell itself
    ctx = hip.get_context()
    up = ctx.upload
    bufs = [up(f['tas']), up(f['tmin']), up(f['rhs']), up(f['wind']), up(f['rsds']), up(f['rlds'])]
    d_pet = pm.run_pmpet_device(ctx, pm.tables_from(d, w.nlcs), w.ncell, 1961, 1970, 0, 6, w.lc_years, *bufs, None,
                                up(w.lct), up(w.elev.reshape(-1)))
    assert np.array_equal(d_pet.download(), got)


@pytest.mark.parametrize('years', [(1899, 1901), (2099, 2101)])
def test_two_leap_year_rules_on_the_device(hip, years):
    """1900 and 2100 are the years where the reference's two calendars disagree: Penman-Monteith takes days per month
    from calendar.isleap (penman_monteith.py:57: February has 28 days), the routing month table from year % 4
    (utils/general.py:37: February has 29 days, 232 sub-steps).  The HIP kernels follow each rule where the reference
    does: PET against the oracle (a 29-day February would change dz and every radiation term), routing bit-exact."""
    from oracle import months as o_months, mrtm as o_mrtm, pm as o_pm
    from xanthos_amd import synth, utils
    from xanthos_amd.pet import penman_monteith as pm
    from xanthos_amd.routing import mrtm
    y0, y1 = years
    w = synth.make_world(nrow=36, ncol=72, ncell=700, n_basins=5, seed=19)
    nm = 36
    f = synth.make_forcing(w, nm)
    d = synth.data_bag(w, f)
    ref = o_pm.run_pmpet(d, w.ncell, w.nlcs, y0, y1, 0, 6, w.lc_years)
    got = pm.run_pmpet(d, w.ncell, w.nlcs, y0, y1, 0, 6, w.lc_years)
    close(got, ref)
    wrong = o_pm.run_pmpet(d, w.ncell, w.nlcs, y0 + 4, y1 + 4, 0, 6, w.lc_years)      # 1904 / 2104 ARE leap years
    feb = 12 + 1
    assert np.max(np.abs(wrong[:, feb] - ref[:, feb]) / (np.abs(ref[:, feb]) + 1e-9)) > 1e-3
    tab = utils.set_month_arrays(nm, y0, y1)
    assert np.array_equal(tab, o_months.set_month_arrays(nm, y0, y1)) and tab[feb, 2] == 29
    st = SimpleNamespace(ngridrow=w.nrow, ngridcol=w.ncol)
    um = mrtm.upstream_genmatrix(mrtm.upstream(w.coords, mrtm.downstream(w.coords, w.flow_dir, st), st))
    runoff = np.random.default_rng(5).gamma(2.0, 30.0, (w.ncell, nm))
    r = o_mrtm.route_series(um.tocsr(), w.flow_dist, w.velocity, w.area, runoff, tab[:, 2], 3)
    g = mrtm.route_series(um, w.flow_dist, w.velocity, w.area, runoff, tab[:, 2], 3, flags=EXACT)
    assert np.array_equal(g[0], r[0]) and np.array_equal(g[1], r[1])
    g = mrtm.route_series(um, w.flow_dist, w.velocity, w.area, runoff, tab[:, 2], 3)              # the default form
    routed_close(g[0], r[0], 1e-3, tag='chs')
    routed_close(g[1], r[1], 1e-9, tag='avg')


COMM_CHILD = r'''
import sys
sys.path.insert(0, sys.argv[1])
import numpy as np
from xanthos_amd import _hip as hip
ctx = hip.get_context(0)
comm = hip.Comm(ctx, 1, 0, hip.comm_unique_id())
rng = np.random.default_rng(0)
n, nm = 300, 40
perm = rng.permutation(n)
local = [ctx.upload(rng.random((n, nm))) for _ in range(3)]
out = [ctx.empty((n, nm)).zero() for _ in range(3)]
comm.gather_rows(local, [n], nm, perm=ctx.upload(perm, dtype=np.int64), out=out, root=0)
ctx.sync()
for a, b in zip(local, out):
    want = np.empty((n, nm))
    want[perm] = a.download()
    assert np.array_equal(b.download(), want)
print('INFO', comm.info()['sends'], comm.info()['recvs'])
comm.close()
print('COMM_OK')
'''


SIDE_GATHER_CHILD = r'''
import sys
sys.path.insert(0, sys.argv[1])
import numpy as np
from xanthos_amd import _hip as hip, synth
from xanthos_amd.pipeline import pipeline_from_world
ctx = hip.get_context(0)
main, side = hip.Comm(ctx, 1, 0, hip.comm_unique_id()), hip.Comm(ctx, 1, 0, hip.comm_unique_id())
w = synth.make_world(nrow=60, ncol=120, ncell=3000, n_basins=7, seed=17, outlet_frac=0.02)
nm = 240
pipe = pipeline_from_world(ctx, w, nm, 1971, 25, 24)
ctx.synth_forcing(23, w.ncell, nm, ctx.upload(w.latitude), pipe.alloc_forcing(), nan_frac=0.004)
pipe.run(fed=False)
ref = pipe.download()
perm_h = np.random.default_rng(1).permutation(w.ncell)
perm = ctx.upload(perm_h, dtype=np.int64)
first, second = ('pet', 'aet', 'q', 'sav'), ('chs', 'avg')
out = {k: ctx.empty((w.ncell, nm)) for k in first + second}
for fed in (True, False, True, True):
    for k in out:
        out[k].zero()
        pipe.out[k].zero()
    n0 = ctx.timing('feed_gate')[1]
    gather_side = lambda: side.gather_rows([pipe.out[k] for k in first], [w.ncell], nm, perm=perm,
                                           out=[out[k] for k in first], root=0, side=True)
    pipe.run(fed=fed, after_runoff=gather_side)        # PET / AET / Q / Sav leave beside the routing kernel
    main.gather_rows([pipe.out[k] for k in second], [w.ncell], nm, perm=perm, out=[out[k] for k in second], root=0)
    ctx.comm_join()
    ctx.sync()
    assert ctx.timing('feed_gate')[1] == n0 + (1 if fed else 0)
    for k in out:
        want = np.empty((w.ncell, nm))
        want[perm_h] = ref[k]
        assert np.array_equal(out[k].download(), want, equal_nan=True), (k, fed)
print('INFO', main.info()['sends'] + side.info()['sends'], main.info()['recvs'] + side.info()['recvs'])
main.close(); side.close()
print('SIDE_OK')
'''


def test_side_gather_is_ordered_behind_the_runoff(hip, tmp_path):
    """xh_comm_gather_rows_side in a FED step: the routing kernel already sits in the context's queue when PET / AET / Q / Sav
    are gathered on the gather stream, so the gather has to order itself behind the side stream that completes the runoff
    (the event a fed xh_run_fused leaves), not behind the context's stream -- and in a stage-by-stage step behind the
    context's stream.  One-rank communicators (real RCCL); outputs zeroed before every pass, so a gather that ran early would
    move zeros."""
    import os
    import subprocess
    import sys
    script = tmp_path / 'side_child.py'
    script.write_text(SIDE_GATHER_CHILD)
    root = os.path.abspath(os.path.join(os.path.dirname(__file__), '..'))
    child = subprocess.Popen([sys.executable, str(script), root], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    try:
        out, _ = child.communicate(timeout=240)
    except subprocess.TimeoutExpired:
        child.kill()
        child.communicate()
        pytest.skip('RCCL communicator bootstrap did not finish within 240 s on this box')
    assert child.returncode == 0 and 'SIDE_OK' in out, out[-3000:]


@pytest.mark.parametrize('child,marker,sends', [('COMM_CHILD', 'COMM_OK', 3), ('SIDE_GATHER_CHILD', 'SIDE_OK', 24)])
def test_real_rccl_send_recv_through_a_self_loop(hip, tmp_path, child, marker, sends):
    """the gather's DATA path through the real librccl.so.1 on a one-GPU box.  RCCL refuses two ranks
    on one device but lets a rank send to itself inside a group; with XH_COMM_SELF_LOOP=1 the root's own rows take the path of
    a remote rank's -- ncclSend (the pipeline's output buffers as send buffers) -> ncclRecv into the staging area -> row
    scatter to grid order -- on the context's stream (the first child) and, in fed and staged steps, on the gather stream
    ordered behind the side stream's runoff event / the context's stream (the second child; outputs zeroed before every
    pass, so a gather that ran early or lost rows would show).  What is still unexercised afterwards: several PEERS
    (ncclCommInitRank across processes, xGMI transport)."""
    import os
    import subprocess
    import sys
    script = tmp_path / 'child.py'
    script.write_text(globals()[child])
    root = os.path.abspath(os.path.join(os.path.dirname(__file__), '..'))
    env = dict(os.environ, XH_COMM_SELF_LOOP='1')
    env.pop('XH_RCCL_LIBRARY', None)                 # the real library, not the test stand-in
    proc = subprocess.Popen([sys.executable, str(script), root], env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    try:
        out, _ = proc.communicate(timeout=300)
    except subprocess.TimeoutExpired:
        proc.kill()
        proc.communicate()
        pytest.skip('RCCL communicator bootstrap did not finish within 300 s on this box')
    assert proc.returncode == 0 and marker in out, out[-3000:]
    info = [ln for ln in out.splitlines() if ln.startswith('INFO ')][-1].split()
    assert int(info[1]) == sends and int(info[2]) == sends, info          # every array went through ncclSend AND ncclRecv


def test_comm_single_rank_gather(hip, tmp_path):
    """The RCCL write-out gather with one rank: librccl is bound at run time, the communicator initialises, and the
    root's own rows go to their grid positions (the send / receive pairs need several GPUs; tests/test_dist_gloo.py
    covers the N > 1 bookkeeping on the CPU).  Runs in a child process with a deadline: RCCL's bootstrap depends on the
    box's network stack, and a stuck bootstrap must not take the rest of the suite with it (it is skipped then)."""
    import os
    import subprocess
    import sys
    script = tmp_path / 'comm_child.py'
    script.write_text(COMM_CHILD)
    root = os.path.abspath(os.path.join(os.path.dirname(__file__), '..'))
    child = subprocess.Popen([sys.executable, str(script), root], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    try:
        out, _ = child.communicate(timeout=180)
    except subprocess.TimeoutExpired:
        child.kill()
        child.communicate()
        pytest.skip('RCCL communicator bootstrap did not finish within 180 s on this box')
    assert child.returncode == 0 and 'COMM_OK' in out, out[-2000:]


@pytest.mark.parametrize('tag', ['snow', 'nosnow'])
def test_abcd_golden(hip, golden, tag):
    from xanthos_amd.runoff import abcd
    g = golden('abcd')
    tmin = g['tmin'] if tag == 'snow' else None
    pet, aet, q, sav = abcd.abcd_execute(int(g['n_basins']), g['basin_ids'], g['pet'], g['precip'], tmin, g['pars'],
                                         int(g['n_months']), int(g['spinup']), jobs=-1)
    assert np.array_equal(pet, g['pet'])
    close(aet, g['aet_' + tag])
    close(q, g['q_' + tag])
    close(sav, g['sav_' + tag])


def test_abcd_class_and_errors(hip, golden):
    from xanthos_amd.runoff import abcd
    g = golden('abcd')
    bid = g['basin_ids']
    he = abcd.ABCD(g['pars'][bid - 1], g['pet'], g['precip'], g['tmin'], bid, int(g['n_months']), int(g['spinup']))
    he.emulate()
    close(he.rsim.T, g['q_snow'])
    close(he.soil_water_storage.T, g['sav_snow'])
    with pytest.raises(IndexError):
        abcd.ABCD(g['pars'][bid - 1], g['pet'], g['precip'], g['tmin'], bid, 60, 24).emulate()
    # odd spin-up length (register-tile tail path)
    from oracle import abcd as o_abcd
    ref = o_abcd.ABCD(g['pars'][bid - 1], g['pet'], g['precip'], g['tmin'], bid, 60, 39)
    ref.emulate()
    he = abcd.ABCD(g['pars'][bid - 1], g['pet'], g['precip'], g['tmin'], bid, 60, 39)
    he.emulate()
    close(he.rsim, ref.rsim)


def _um(t, tag):
    from xanthos_amd.routing import mrtm
    return mrtm.UpstreamMatrix(t[tag + '_um_indptr'], t[tag + '_um_indices'], t[tag + '_um_data'])


@pytest.mark.parametrize('flags', [0, 1])
@pytest.mark.parametrize('tag', ['rand', 'tree'])
def test_streamrouting_golden_bit_exact(hip, golden, tag, flags):
    from xanthos_amd.routing import mrtm
    g, t = golden('mrtm'), golden('topo')
    um = _um(t, tag)
    S = g[tag + '_S0']
    n = len(S)
    for nday in (28, 29, 30, 31):
        S, favg, F = mrtm.streamrouting(g[tag + '_L'], S, np.zeros(n), g[tag + '_chv'], g['%s_q_%d' % (tag, nday)],
                                        g[tag + '_area'], nday, 10800, um, flags=flags | EXACT)
        assert np.array_equal(S, g['%s_S_%d' % (tag, nday)])
        assert np.array_equal(favg, g['%s_Favg_%d' % (tag, nday)])
        assert np.array_equal(F, g['%s_F_%d' % (tag, nday)])


@pytest.mark.parametrize('flags', [0, 8, 4])    # time-skewed units (default) / lock-step units / one workgroup per network
@pytest.mark.parametrize('tag', ['rand', 'tree'])
def test_route_series_golden_bit_exact(hip, golden, tag, flags):
    from xanthos_amd.routing import mrtm
    g, t = golden('mrtm'), golden('topo')
    chs, avg, fend = mrtm.route_series(_um(t, tag), g[tag + '_L'], g[tag + '_chv'], g[tag + '_area'],
                                       g[tag + '_series_runoff'], g[tag + '_series_ndays'], int(g['series_spinup']),
                                       flags=flags | EXACT)
    assert np.array_equal(chs, g[tag + '_series_chstorage'])
    assert np.array_equal(avg, g[tag + '_series_avgchflow'])
    assert np.array_equal(fend, g[tag + '_series_Fend'])


def test_route_series_atomic_variant_close(hip, golden):
    """global_atomic_add_f64 scatter variant: same physics, summation order not reproducible -> tolerance."""
    from xanthos_amd.routing import mrtm
    g, t = golden('mrtm'), golden('topo')
    tag = 'tree'
    chs, avg, _ = mrtm.route_series(_um(t, tag), g[tag + '_L'], g[tag + '_chv'], g[tag + '_area'],
                                    g[tag + '_series_runoff'], g[tag + '_series_ndays'], int(g['series_spinup']),
                                    flags=3)
    close(avg, g[tag + '_series_avgchflow'], rtol=1e-9, atol=1e-6)
    close(chs, g[tag + '_series_chstorage'], rtol=1e-9, atol=1e-3)


@pytest.mark.parametrize('flags', [0, 8, 4])
def test_route_synthetic_world_vs_oracle(hip, flags):
    """A 3000-cell world with networks far larger than one unit, 14 months incl. spin-up: bit-exact both as
    dataflow units linked by streams (flags=0) and as one workgroup per network (flags=4)."""
    from types import SimpleNamespace as NS
    from oracle import months as o_months
    from oracle import mrtm as o_mrtm
    from xanthos_amd import synth
    from xanthos_amd.routing import mrtm
    w = synth.make_world(nrow=60, ncol=120, ncell=3000, n_basins=5, seed=3, outlet_frac=0.02)
    st = NS(ngridrow=w.nrow, ngridcol=w.ncol)
    ds = mrtm.downstream(w.coords, w.flow_dir, st)
    um = mrtm.upstream_genmatrix(mrtm.upstream(w.coords, ds, st))
    info = um.plan(hip.get_context()).info()
    assert info['largest_network'] > 256 and info['fallback_cells'] == 0 and info['units'] < info['networks']
    assert info['flow_cells'] == w.ncell and info['flow_edges'] > 10 and info['flow_depth'] > 2
    rng = np.random.default_rng(9)
    runoff = rng.gamma(2.0, 30.0, (w.ncell, 12))
    runoff[rng.random(w.ncell) < 0.01] = np.nan          # NaN runoff (NaN precip cells) must propagate identically
    ndays = o_months.set_month_arrays(12, 1972, 1972)[:, 2]
    ref = o_mrtm.route_series(um.tocsr(), w.flow_dist, w.velocity, w.area, runoff, ndays, 2)
    got = mrtm.route_series(um, w.flow_dist, w.velocity, w.area, runoff, ndays, 2, flags=flags | EXACT)
    for a, b in zip(got, ref):
        assert np.array_equal(a, b, equal_nan=True)
    assert um.plan(hip.get_context()).info()['last_tree_kernel'] == {0: 2, 8: 1, 4: 0}[flags]


_PARTITION_CHILD = r"""
import sys, json
import numpy as np
sys.path.insert(0, sys.argv[1])
from types import SimpleNamespace as NS
from oracle import months as o_months
from oracle import mrtm as o_mrtm
from xanthos_amd import _hip, synth
from xanthos_amd.routing import mrtm
w = synth.make_world(nrow=60, ncol=120, ncell=3000, n_basins=5, seed=3, outlet_frac=0.02)
st = NS(ngridrow=w.nrow, ngridcol=w.ncol)
ds = mrtm.downstream(w.coords, w.flow_dir, st)
um = mrtm.upstream_genmatrix(mrtm.upstream(w.coords, ds, st))
rng = np.random.default_rng(11)
runoff = rng.gamma(2.0, 30.0, (w.ncell, 12))
ndays = o_months.set_month_arrays(12, 1973, 1973)[:, 2]
ref = o_mrtm.route_series(um.tocsr(), w.flow_dist, w.velocity, w.area, runoff, ndays, 2)
import os
ok = True
for rep in range(int(os.environ.get('XH_TEST_CALLS', '1'))):
    got = mrtm.route_series(um, w.flow_dist, w.velocity, w.area, runoff, ndays, 2, flags=int(os.environ.get('XH_TEST_FLAGS', '0')) | 256)      # (256 = XH_ROUTE_EXACT)
    ok = ok and all(np.array_equal(a, b, equal_nan=True) for a, b in zip(got, ref))
plan = um.plan(_hip.get_context())
info = plan.info()
print(json.dumps({'ok': bool(ok), 'kernel': int(info['last_tree_kernel']), 'units': int(info['flow_units']),
                  'edges': int(info['flow_edges']), 'reroutes': int(info['reroutes'])}))
"""


@pytest.mark.parametrize('env', [{}, {'XH_FLOW_PIECE_CAP': '64'}, {'XH_FLOW_PIECE_CAP': '20'}, {'XH_FLOW_SPARE': '0'},
                                 {'XH_FLOW_SPARE': '700'}, {'XH_FLOW_RS': '16384'}, {'XH_TEST_CALLS': '4'},
                                 {'XH_TEST_FLAGS': '8'}])
def test_route_partition_variants_bit_exact(env, tmp_path):
    """The knobs of the bit-exact kernel's dataflow partition (piece capacity, spare workgroups, ring size; repeated calls on one
    plan; the lock-step kernel, flags 8) change which cells share a wave and who waits for whom -- never a bit of the result.
    Each variant routes the 3000-cell world in a process of its own (the knobs are read once per process) against the oracle."""
    import json
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    script = tmp_path / 'child.py'
    script.write_text(_PARTITION_CHILD)
    e = dict(os.environ)
    e.update(env)
    out = subprocess.run([sys.executable, str(script), root], env=e, stdout=subprocess.PIPE, stderr=subprocess.PIPE,
                         text=True, timeout=300)
    assert out.returncode == 0, out.stderr[-2000:]
    res = json.loads(out.stdout.strip().splitlines()[-1])
    assert res['ok'] and res['kernel'] == (1 if env.get('XH_TEST_FLAGS') == '8' else 2) and res['reroutes'] == 0, res
    if env.get('XH_FLOW_PIECE_CAP') == '20':
        assert res['edges'] > 150, res          # many more streams than the default cut


@pytest.mark.parametrize('limit,kernel', [(3000 * 12 * 8, 2), (3000 * 12 * 8 - 1, 1)])
def test_route_row_offset_limit(limit, kernel, tmp_path):
    """k_mrtm_wave addresses a cell's runoff row with a 32-bit byte offset: a grid whose rows reach 4 GiB must be routed
    by a kernel with 64-bit offsets instead of wrapping silently.  XH_WAVE_ROW_LIMIT moves the limit down to the 3000-cell x
    12-month world: exactly at the limit the time-skewed kernel routes it, one byte below the lock-step kernel does --
    bit-exact either way."""
    import json
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    script = tmp_path / 'child.py'
    script.write_text(_PARTITION_CHILD)
    e = dict(os.environ)
    e['XH_WAVE_ROW_LIMIT'] = str(limit)
    out = subprocess.run([sys.executable, str(script), root], env=e, stdout=subprocess.PIPE, stderr=subprocess.PIPE,
                         text=True, timeout=300)
    assert out.returncode == 0, out.stderr[-2000:]
    res = json.loads(out.stdout.strip().splitlines()[-1])
    assert res['ok'] and res['kernel'] == kernel and res['reroutes'] == 0, res


@pytest.mark.parametrize('basin', [0, 1])
@pytest.mark.parametrize('unit', ['km3_per_mth', 'mm_per_mth'])
@pytest.mark.parametrize('tag', ['snow', 'nosnow'])
def test_calibration_objective_golden(hip, golden, basin, unit, tag):
    g = golden('kge')
    b = str(basin)
    ctx = hip.get_context()
    nm, spin = int(g['n_months']), int(g['spinup'])
    npar = 5 if tag == 'snow' else 4
    tr = lambda a: ctx.upload(np.ascontiguousarray(a.T))
    d_tmin = tr(g['tmin_' + b]) if tag == 'snow' else None
    d_area = ctx.upload(g['areas_' + b]) if unit == 'km3_per_mth' else None
    ed, series = ctx.calib_objective(g['pet_' + b].shape[0], nm, spin, g['pars_' + b][:, :npar], tr(g['pet_' + b]),
                                     tr(g['precip_' + b]), d_tmin, d_area, g['robs_' + b], want_series=True)
    close(series, g['series_%s_%s_%s' % (b, unit, tag)], rtol=1e-9, atol=1e-12)
    close(ed, g['ed_%s_%s_%s' % (b, unit, tag)], rtol=1e-9, atol=1e-12)


def test_edge_sizes(hip):
    """Smallest legal problems: one cell, one year; a two-cell network; outputs that are not a multiple of 8 months."""
    from types import SimpleNamespace as NS
    from oracle import abcd as o_abcd, mrtm as o_mrtm, pm as o_pm
    from xanthos_amd import synth
    from xanthos_amd.pet import penman_monteith as pm
    from xanthos_amd.routing import mrtm
    from xanthos_amd.runoff import abcd
    w = synth.make_world(nrow=12, ncol=24, ncell=40, n_basins=2, seed=4)
    f = synth.make_forcing(w, 36, nan_precip=False)
    one = NS(**{k: getattr(w, k) for k in ('cL', 'beta', 'rslimit', 'ae', 'be', 'Tminopen', 'Tminclose', 'VPDclose',
                                           'VPDopen', 'RBLmin', 'RBLmax', 'rc', 'emiss', 'alpha', 'lai', 'laimax',
                                           'laimin')})
    one.elev, one.lct = w.elev[:1], w.lct[:1]
    d = synth.data_bag(one, {k: v[:1, :12] for k, v in f.items()})
    close(pm.run_pmpet(d, 1, w.nlcs, 1980, 1980, 0, 6, w.lc_years), o_pm.run_pmpet(d, 1, w.nlcs, 1980, 1980, 0, 6, w.lc_years))
    pars = w.abcd_pars[:1]
    got = abcd.abcd_execute(1, np.array([1]), f['rsds'][:1, :26] * 0.5, f['precip'][:1, :26], f['abcd_tmin'][:1, :26], pars, 26, 25)
    ref = o_abcd.abcd_execute(1, np.array([1]), f['rsds'][:1, :26] * 0.5, f['precip'][:1, :26], f['abcd_tmin'][:1, :26], pars, 26, 25, 1)
    for a, b in zip(got[1:], ref[1:]):
        close(a, b)
    # two cells, cell 1 drains into cell 0; 13 months (1 group of 8 + partial group of 5), with and without spin-up
    um = mrtm.UpstreamMatrix([0, 2, 3], [0, 1, 1], [-1, 1, -1])
    L, v, area = np.array([30e3, 1000.0]), np.array([1.0, 2.0]), np.array([2500.0, 2400.0])
    q = np.random.default_rng(2).gamma(2.0, 30.0, (2, 13))
    ndays = np.array([31, 28, 31, 30, 31, 30, 31, 31, 30, 31, 30, 31, 31])
    for spin in (0, 5):
        ref = o_mrtm.route_series(um.tocsr(), L, v, area, q, ndays, spin)
        for flags in (0, 8, 4, 1):
            got = mrtm.route_series(um, L, v, area, q, ndays, spin, flags=flags | EXACT)
            for a, b in zip(got, ref):
                assert np.array_equal(a, b), (spin, flags)
        got = mrtm.route_series(um, L, v, area, q, ndays, spin)      # the library's default form
        for a, b, atol in zip(got, ref, (1e-3, 1e-9, 1e-9)):
            routed_close(a, b, atol, tag=('default form', spin))


def test_argument_errors_are_reported(hip):
    from xanthos_amd.routing import mrtm
    from xanthos_amd.runoff import abcd
    ctx = hip.get_context()
    with pytest.raises(hip.HipError):                                   # nmonths not a multiple of 12
        ctx.pm_pet({k: np.ones(8 * (12 if k in ('alpha', 'lai', 'laimin', 'laimax') else 1)) for k in
                    ('cL', 'beta', 'rslimit', 'Tminopen', 'Tminclose', 'VPDclose', 'VPDopen', 'RBLmin', 'RBLmax', 'rc',
                     'emiss', 'alpha', 'lai', 'laimin', 'laimax')}, 4, 13, 2000, [2000], 0, 6,
                   *[ctx.empty((4, 13)) for _ in range(6)], None, ctx.empty((4, 8, 1)), ctx.empty(4), ctx.empty((4, 13)))
    with pytest.raises(IndexError):                                      # spin-up shorter than 25 months
        abcd.abcd_execute(1, np.ones(3, dtype=int), np.ones((3, 36)), np.ones((3, 36)), None, np.ones((1, 5)) * 0.5, 36, 12)
    um = mrtm.UpstreamMatrix([0, 1], [0], [-1])
    with pytest.raises(hip.HipError):                                    # spin-up longer than the series
        mrtm.route_series(um, [1e4], [1.0], [1e3], np.ones((1, 3)), [30, 30, 30], 5)
    with pytest.raises(ValueError):
        mrtm.UpstreamMatrix.from_scipy(__import__('scipy.sparse').sparse.csr_matrix(np.array([[2.0]])))


def test_writer_aggregation_golden(hip, golden, tmp_path):
    """Device-side month -> year aggregation, mm -> km3 and spatial sums against the reference's OutWriter."""
    from types import SimpleNamespace as NS
    from xanthos_amd.data_writer.out_writer import OutWriter
    g = golden('writer')
    q = g['q']
    s = NS(output_vars=['q', 'avgchflow'], ProjectName='p', OutputFolder=str(tmp_path), OutputFormat=4, OutputUnit=1,
           OutputInYear=1, StartYear=2001, EndYear=2003, device=0)
    w = OutWriter(s, g['area'], {'q': q, 'avgchflow': q})
    close(w.agg_to_year(q, 'sum'), g['ysum'], rtol=1e-12, atol=0)
    close(w.agg_to_year(q, 'mean'), g['ymean'], rtol=1e-12, atol=0)
    close(w._agg(q, 1, 0, g['area'] / 1e6), g['km3'], rtol=1e-14, atol=0)
    w.write()
    close(w.get('q'), g['ysum_km3'], rtol=1e-12, atol=0)              # yearly sum, then x area / 1e6
    close(w.get('avgchflow'), g['ymean'], rtol=1e-12, atol=0)         # channel flow: yearly mean, no conversion
    assert np.array_equal(np.load(str(tmp_path / 'q_km3peryear_p.npy')), w.get('q'), equal_nan=True)
    close(w.agg_spatial(w.get('q'), g['ids'], 8), g['spatial'], rtol=1e-12, atol=0)
    # a device-resident input gives the same result without the upload
    d_q = hip.get_context().upload(q)
    w2 = OutWriter(s, g['area'], {'q': d_q})
    w2.write()
    assert np.array_equal(w2.get('q'), w.get('q'), equal_nan=True)


@pytest.mark.parametrize('dt', [7200, 17280, 86400])
def test_route_other_time_steps(hip, dt):
    """Sub-step counts that are not a multiple of the 8-step stream blocks, streams between units; with one sub-step a
    day the months are shorter than the deepest lane lag and the default path must fall back to lock-step units."""
    from types import SimpleNamespace as NS
    from oracle import mrtm as o_mrtm
    from xanthos_amd import synth
    from xanthos_amd.routing import mrtm
    w = synth.make_world(nrow=60, ncol=120, ncell=3000, n_basins=5, seed=3, outlet_frac=0.02)
    st = NS(ngridrow=w.nrow, ngridcol=w.ncol)
    um = mrtm.upstream_genmatrix(mrtm.upstream(w.coords, mrtm.downstream(w.coords, w.flow_dir, st), st))
    runoff = np.random.default_rng(12).gamma(2.0, 30.0, (w.ncell, 5))
    ndays = np.array([31, 28, 31, 30, 31])
    assert any(int(d * 86400 / dt) % 8 for d in ndays)
    ref = o_mrtm.route_series(um.tocsr(), w.flow_dist, w.velocity, w.area, runoff, ndays, 2, dt=dt)
    plan = um.plan(hip.get_context())
    for flags in (0, 8, 4):
        got = mrtm.route_series(um, w.flow_dist, w.velocity, w.area, runoff, ndays, 2, dt=dt, flags=flags | EXACT)
        for a, b in zip(got, ref):
            assert np.array_equal(a, b), (dt, flags)
        info = plan.info()
        short = min(int(d * 86400 / dt) for d in ndays) < info['skew_max_lag'] + 32
        assert info['last_tree_kernel'] == {0: 1 if short else 2, 8: 1, 4: 0}[flags], (dt, flags, info)


def test_calibration_objective_multi_basin(hip, golden):
    """Both golden basins in ONE launch, each with its own population: identical to the per-basin calls."""
    g = golden('kge')
    ctx = hip.get_context()
    nm, spin = int(g['n_months']), int(g['spinup'])
    tr = lambda a: ctx.upload(np.ascontiguousarray(a.T))
    for npar, with_tmin in ((5, True), (4, False)):
        pars = np.stack([g['pars_0'][:, :npar], g['pars_1'][::-1, :npar]])
        ncells = [g['pet_0'].shape[0], g['pet_1'].shape[0]]
        pet = [tr(g['pet_0']), tr(g['pet_1'])]
        pr = [tr(g['precip_0']), tr(g['precip_1'])]
        tn = [tr(g['tmin_0']), tr(g['tmin_1'])] if with_tmin else None
        ar = [ctx.upload(g['areas_0']), ctx.upload(g['areas_1'])]
        obs = np.stack([g['robs_0'], g['robs_1']])
        ed, series = ctx.calib_objective_multi(ncells, nm, spin, pars, pet, pr, tn, ar, obs, want_series=True)
        for b in range(2):
            one, s1 = ctx.calib_objective(ncells[b], nm, spin, pars[b], pet[b], pr[b], tn[b] if tn else None, ar[b],
                                          obs[b], want_series=True)
            assert np.array_equal(ed[b], one) and np.array_equal(series[b], s1)
        tag = 'snow' if with_tmin else 'nosnow'
        close(ed[0], g['ed_0_km3_per_mth_' + tag], rtol=1e-9, atol=1e-12)
        close(ed[1], g['ed_1_km3_per_mth_' + tag][::-1], rtol=1e-9, atol=1e-12)


@pytest.mark.parametrize('flags', [0, 8])
def test_route_long_chain_deep_lags(hip, flags):
    """A 1,500-cell main stem with short side branches and cells that fire every other sub-step: pieces are 64-cell
    chains, so lane lags reach the maximum (128 sub-steps) and every month boundary is crossed lane by lane; initial
    storage given, 11 months (one 8-month output group + a partial one), spin-up 3."""
    from oracle import mrtm as o_mrtm
    from xanthos_amd.routing import mrtm
    rng = np.random.default_rng(77)
    n_main, n = 1500, 1500 + 300
    ds = np.full(n, -1)
    ds[1:n_main] = np.arange(0, n_main - 1)                       # cell k drains to k - 1; cell 0 is the outlet
    ds[n_main:] = rng.integers(5, n_main, n - n_main)             # side cells drain into the main stem
    perm = rng.permutation(n)                                     # scramble the ids: row order != river order
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
    info = um.plan(hip.get_context()).info()
    assert info['flow_cells'] == n and info['skew_max_lag'] >= 112 and info['flow_depth'] >= 10
    L = rng.uniform(20e3, 60e3, n)
    L[rng.random(n) < 0.03] = 4e3                                  # velocity x dt > length: these cells fire
    v = rng.uniform(0.4, 1.5, n)
    area = rng.uniform(800, 3100, n)
    q = rng.gamma(2.0, 30.0, (n, 11))
    S0 = rng.uniform(0.0, 5e7, n)
    ndays = np.array([31, 28, 31, 30, 31, 30, 31, 31, 30, 31, 30])
    ref = o_mrtm.route_series(um.tocsr(), L, v, area, q, ndays, 3, S0=S0)
    got = mrtm.route_series(um, L, v, area, q, ndays, 3, S0=S0, flags=flags | EXACT)
    for a, b in zip(got, ref):
        assert np.array_equal(a, b)
    assert um.plan(hip.get_context()).info()['last_tree_kernel'] == (2 if flags == 0 else 1)


def test_route_differential_fuzz(hip):
    """tools/fuzz_routing.py, 25 seeded cases: random tree worlds (12 .. 6,000 cells), month counts, spin-ups, time steps,
    NaN runoff, stagnant and over-fast channels, initial storage -- all three network kernels bit-exact against the oracle."""
    import importlib.util
    import os
    spec = importlib.util.spec_from_file_location('fuzz_routing', os.path.join(os.path.dirname(__file__), '..', 'tools', 'fuzz_routing.py'))
    fz = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(fz)
    rng = np.random.default_rng(31337)
    used = [fz.one_case(rng, k)[4][0] for k in range(25)]
    assert 2 in used                      # the time-skewed kernel was exercised


def test_pm_abcd_parity_fuzz(hip):
    """tools/fuzz_pm_abcd.py, 10 seeded cases with hostile forcing (zeros from nan_to_num, saturated air, -60 K spikes, NaN
    precipitation, with / without snow): PET / AET / Q / Sav within 1 % of the 1e-6 north-star gate."""
    import importlib.util
    import os
    spec = importlib.util.spec_from_file_location('fuzz_pm_abcd', os.path.join(os.path.dirname(__file__), '..', 'tools', 'fuzz_pm_abcd.py'))
    fz = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(fz)
    rng = np.random.default_rng(4242)
    worst = max(max(fz.one_case(rng, k)[3]) for k in range(10))
    assert worst < 1e-2, worst


@pytest.mark.parametrize('nm,abcd_spin,route_spin', [(240, 25, 24), (120, 36, 12), (600, 120, 120)])
def test_fed_routing_equals_stage_by_stage(hip, nm, abcd_spin, route_spin):
    """xh_run_fused mode 1 (components.py:344-370 is the hand-over it replaces): the routing kernel starts once the first
    max(spin-ups) months of runoff exist and is fed the remaining months, produced beside it on a second stream, through the
    months-ready word.  Same kernels, same arithmetic: all six outputs bit-identical to the stages run one after the other,
    call after call (the staged runoff, the words and the stream rings are reused), NaN-precipitation cells included.
    Runs on what the library ships: the reassociated routing form on the pipeline's PREPARED plan (single running sums)."""
    from xanthos_amd import synth
    from xanthos_amd.pipeline import OUTPUTS, pipeline_from_world
    ctx = hip.get_context()
    w = synth.make_world(nrow=60, ncol=120, ncell=3000, n_basins=7, seed=17, outlet_frac=0.02)
    pipe = pipeline_from_world(ctx, w, nm, 1971, abcd_spin, route_spin)
    lat = ctx.upload(w.latitude)
    # Two forcings, alternating: the staged copy of the runoff, the hand-over words and the stream rings are
    # reused from call to call, so with ONE forcing a stale staged line -- left in a cache by the previous call -- would hold
    # the right values and pass.  With two, every call must read what ITS OWN side stream wrote.
    refs = {}
    for seed in (23, 57):
        ctx.synth_forcing(seed, w.ncell, nm, lat, pipe.alloc_forcing(), nan_frac=0.004)
        pipe.run(fed=False, fused=False)
        refs[seed] = pipe.download()
        ref = refs[seed]
        assert np.isnan(ref['q']).any() and np.isfinite(ref['avg']).any() and np.nanmax(ref['chs']) > 0
    assert not np.array_equal(refs[23]['q'], refs[57]['q'], equal_nan=True)
    n0 = ctx.timing('feed_gate')[1]
    for rep in range(4):
        seed = (23, 57)[rep & 1]
        ctx.synth_forcing(seed, w.ncell, nm, lat, pipe.alloc_forcing(), nan_frac=0.004)
        for k in OUTPUTS:
            pipe.out[k].zero()
        pipe.run(fed=True, fused=False)
        got = pipe.download()
        for k in OUTPUTS:
            assert np.array_equal(got[k], refs[seed][k], equal_nan=True), (k, rep)
    ref = refs[57]
    # the calls really were routed that way (the gate kernel of the side stream ran once per call), without a re-route
    assert ctx.timing('feed_gate')[1] == n0 + 4
    assert pipe.plan.info()['reroutes'] == 0 and pipe.plan.info()['last_tree_kernel'] == 4
    ri = pipe.plan.rsum_info()
    assert ri['pair_cells'] >= 0 and ri['fold_disabled'] == 0, ri            # the prepared plan: single sums, no guard trip
    # flags the fed call cannot take fall back to the stage-by-stage order inside the same entry point: same results (routed
    # by the workgroup-per-network kernel then, which is bit-exact: within the default form's bar of the runs above)
    pipe.route_flags = hip.XH_ROUTE_NO_DATAFLOW
    for k in OUTPUTS:
        pipe.out[k].zero()
    pipe.run(fed=True, fused=False)
    got = pipe.download()
    outputs_close(got, ref, OUTPUTS, 'no dataflow')
    assert ctx.timing('feed_gate')[1] == n0 + 4 and pipe.plan.info()['last_tree_kernel'] == 0


def test_fault_inside_a_fed_call_is_settled_silently(hip):
    """a routing fault inside a FED call (XH_ROUTE_TEST_FAULT raises the fault word as a timed-out wait would)
    is settled like one inside a stage-by-stage call -- the call is routed again from the complete runoff array and the
    caller hears nothing: what the side stream enqueued behind the routing kernel (the rest of PM and ABCD) reads none of
    the routing's outputs and must not count as "work that consumed invalid results"."""
    from xanthos_amd import synth
    from xanthos_amd.pipeline import OUTPUTS, pipeline_from_world
    ctx = hip.get_context()
    w = synth.make_world(nrow=60, ncol=120, ncell=3000, n_basins=7, seed=19, outlet_frac=0.02)
    nm = 240
    pipe = pipeline_from_world(ctx, w, nm, 1971, 25, 24)
    ctx.synth_forcing(41, w.ncell, nm, ctx.upload(w.latitude), pipe.alloc_forcing(), nan_frac=0.002)
    pipe.run(fed=False, fused=False)
    ref = pipe.download()
    assert pipe.plan.info()['last_tree_kernel'] == 4                # the library's default form, on the prepared plan
    r0, n0 = pipe.plan.info()['reroutes'], ctx.timing('feed_gate')[1]
    pipe.route_flags = hip.XH_ROUTE_TEST_FAULT
    for k in OUTPUTS:
        pipe.out[k].zero()
    pipe.run(fed=True, fused=False)
    ctx.sync()                                   # settles the fault: no exception
    got = pipe.download()
    outputs_close(got, ref, OUTPUTS, 'after the fault')      # (routed again by the bit-exact workgroup-per-network kernel)
    assert ctx.timing('feed_gate')[1] == n0 + 1                  # the call really ran in the fed order ...
    assert pipe.plan.info()['reroutes'] == r0 + 1                # ... and its routing was done again after the fault
    # the same fault followed by work that DOES read the routing's outputs is still reported
    rows = ctx.upload(np.arange(8, dtype=np.int64), dtype=np.int64)      # (made first: a synchronous upload would settle the fault)
    tmp = ctx.empty((8, nm))
    pipe.route_flags = hip.XH_ROUTE_TEST_FAULT
    pipe.run(fed=True, fused=False)
    ctx.gather_rows(pipe.out['chs'], rows, 8, nm, tmp)
    with pytest.raises(hip.HipError):
        ctx.sync()
    pipe.route_flags = 0
    # (the plan backs off after a fault; the next calls route without the dataflow kernels and still agree)
    pipe.run(fed=True, fused=False)
    got = pipe.download()
    outputs_close(got, ref, OUTPUTS, 'backed off')


def test_prepared_plan_notices_new_velocities_in_place(hip):
    """The default form's counterpart: the pipeline's prepared plan (single running sums, folded leaves) was made for the
    cells that can fire at the velocities it was given.  Velocities overwritten IN PLACE so that other cells can fire are
    noticed by the kernel's guard (a cell that can fire and is not marked), the call is routed again on the plan of pairs --
    staged and fed -- and every routed value stays within the bar of the oracle on the new velocities."""
    from oracle import mrtm as o_mrtm
    from xanthos_amd import synth
    from xanthos_amd.pipeline import pipeline_from_world
    ctx = hip.get_context()
    w = synth.make_world(nrow=60, ncol=120, ncell=3000, n_basins=7, seed=29, outlet_frac=0.02)
    nm = 48
    pipe = pipeline_from_world(ctx, w, nm, 1971, 25, 12)
    ctx.synth_forcing(31, w.ncell, nm, ctx.upload(w.latitude), pipe.alloc_forcing(), nan_frac=0.002)
    pipe.run(fed=False)
    q = pipe.out['q'].download()
    ri = pipe.plan.rsum_info()
    assert pipe.plan.info()['last_tree_kernel'] == 4 and ri['pair_cells'] >= 0 and ri['fold_disabled'] == 0, ri
    ref = o_mrtm.route_series(pipe.um.tocsr(), w.flow_dist, w.velocity, w.area, q, pipe.ndays, 12)
    routed_close(pipe.out['chs'].download(), ref[0], 1e-3, tag='first velocities, chs')
    routed_close(pipe.out['avg'].download(), ref[1], 1e-9, tag='first velocities, avg')
    rng = np.random.default_rng(3)
    ratio = w.velocity * 10800.0 / w.flow_dist
    v2 = w.velocity.copy()
    flip = rng.random(w.ncell) < 0.2
    v2[flip & (ratio <= 1.0)] *= 3.0 / np.maximum(ratio[flip & (ratio <= 1.0)], 0.05)
    pipe.d_velocity.upload(v2)
    ref = o_mrtm.route_series(pipe.um.tocsr(), w.flow_dist, v2, w.area, q, pipe.ndays, 12)
    for fed in (False, True):
        for k in ('chs', 'avg'):
            pipe.out[k].zero()
        pipe.run(fed=fed)
        routed_close(pipe.out['chs'].download(), ref[0], 1e-3, tag=('new velocities, chs', fed))
        routed_close(pipe.out['avg'].download(), ref[1], 1e-9, tag=('new velocities, avg', fed))
        ri = pipe.plan.rsum_info()
        assert pipe.plan.info()['last_tree_kernel'] == 4 and ri['fold_disabled'] == 1 and ri['pair_cells'] == -1, (fed, ri)


def test_first_dataflow_call_of_a_plan_is_cross_checked(hip, tmp_path, monkeypatch):
    """The dataflow kernels' streams rest on an ordering assumption outside the HIP memory model (xh_mrtm_wave.hip, check();
    the fenced form costs +86 %, profiles/round4/fenced_ab.txt).  So the FIRST dataflow call of a plan on a device / library
    build without a pass on record is routed again by the barrier-only kernel and compared on the device; the pass is
    recorded under XH_CACHE_DIR and later plans of the same topology skip the check."""
    from xanthos_amd import synth
    from xanthos_amd.routing import mrtm
    from oracle import months as o_months, mrtm as o_mrtm
    monkeypatch.setenv('XH_CACHE_DIR', str(tmp_path / 'cache'))
    w = synth.make_world(nrow=60, ncol=120, ncell=2500, n_basins=5, seed=77, outlet_frac=0.02)
    st = SimpleNamespace(ngridrow=w.nrow, ngridcol=w.ncol)
    ds = mrtm.downstream(w.coords, w.flow_dir, st)
    rng = np.random.default_rng(5)
    runoff = rng.gamma(2.0, 30.0, (w.ncell, 12))
    runoff[rng.random(w.ncell) < 0.01] = np.nan
    ndays = o_months.set_month_arrays(12, 1975, 1975)[:, 2]

    def fresh():
        return mrtm.upstream_genmatrix(mrtm.upstream(w.coords, ds, st))      # a new UpstreamMatrix = a new plan
    um = fresh()
    ref = o_mrtm.route_series(um.tocsr(), w.flow_dist, w.velocity, w.area, runoff, ndays, 2)
    def held(got):
        for a, b, atol in zip(got, ref, (1e-3, 1e-9, 1e-9)):      # the library's default form: within its bar
            routed_close(a, b, atol)
    for call, want in ((0, 1), (1, 1), (2, 1)):
        got = mrtm.route_series(um, w.flow_dist, w.velocity, w.area, runoff, ndays, 2)
        held(got)
        assert um.plan(hip.get_context()).info()['validated'] == want, call
    assert um.plan(hip.get_context()).info()['last_tree_kernel'] == 4
    marks = [f for f in (tmp_path / 'cache').iterdir() if f.name.startswith('route_ok_')]
    assert len(marks) == 1
    um2 = fresh()                                                             # same topology, same box, same build: on record
    mrtm.route_series(um2, w.flow_dist, w.velocity, w.area, runoff, ndays, 2)
    assert um2.plan(hip.get_context()).info()['validated'] == 0
    monkeypatch.setenv('XH_CACHE_DIR', str(tmp_path / 'elsewhere'))           # no record there: checked again
    um3 = fresh()
    mrtm.route_series(um3, w.flow_dist, w.velocity, w.area, runoff, ndays, 2)
    assert um3.plan(hip.get_context()).info()['validated'] == 1
    # round 5: the record is also keyed on the HIP runtime and driver versions (the ordering assumption is theirs as much as
    # the silicon's): a pass recorded under another runtime does not count
    monkeypatch.setenv('XH_CACHE_DIR', str(tmp_path / 'cache'))
    monkeypatch.setenv('XH_TEST_RUNTIME_TAG', 'another runtime')
    um4 = fresh()
    mrtm.route_series(um4, w.flow_dist, w.velocity, w.area, runoff, ndays, 2)
    assert um4.plan(hip.get_context()).info()['validated'] == 1
    monkeypatch.delenv('XH_TEST_RUNTIME_TAG')
    # ... and a long-lived plan is cross-checked again every XH_ROUTE_VALIDATE_EVERY-th dataflow call (default 1,000)
    monkeypatch.setenv('XH_ROUTE_VALIDATE_EVERY', '3')
    um5 = fresh()
    seen = []
    for call in range(7):
        got = mrtm.route_series(um5, w.flow_dist, w.velocity, w.area, runoff, ndays, 2)
        held(got)
        seen.append(um5.plan(hip.get_context()).info()['validated'])
    assert seen == [0, 0, 1, 1, 1, 2, 2], seen
