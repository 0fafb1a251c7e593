"""The oracle (numpy restatement) against the committed golden vectors made from the real reference.

tests/golden/make_golden.py imported JGCRI/xanthos v2.4.1 in the build container and stored crafted inputs with
the reference's outputs.  These tests pin every oracle function to those outputs (CPU only).
"""
from types import SimpleNamespace

import numpy as np
import pytest

from oracle import abcd as o_abcd
from oracle import calib as o_calib
from oracle import months as o_months
from oracle import mrtm as o_mrtm
from oracle import pm as o_pm

PM_FIELDS = ('cL', 'beta', 'rslimit', 'ae', 'be', 'Tminopen', 'Tminclose', 'VPDclose', 'VPDopen', 'RBLmin', 'RBLmax',
             'rc', 'emiss', 'alpha', 'lai', 'laimax', 'laimin', 'elev', 'tair_load', 'TMIN_load', 'rhs_load',
             'wind_load', 'rsds_load', 'rlds_load', 'tairprev_load', 'lct_load')


def pm_bag(g):
    return SimpleNamespace(**{k: g[k] for k in PM_FIELDS})


def rel_err(x, ref):
    x, ref = np.asarray(x, float), np.asarray(ref, float)
    assert np.array_equal(np.isnan(x), np.isnan(ref))
    m = ~np.isnan(ref)
    return float(np.max(np.abs(x[m] - ref[m]) / (np.abs(ref[m]) + 1e-300))) if m.any() else 0.0


def test_pm_matches_reference(golden):
    g = golden('pm')
    d = pm_bag(g)
    n = d.tair_load.shape[0]
    lcy = [int(v) for v in g['lc_years']]
    pet = o_pm.run_pmpet(d, n, int(g['nlcs']), int(g['start_year']), int(g['end_year']), 0, 6, lcy)
    assert pet.shape == g['pet'].shape
    assert np.allclose(pet, g['pet'], rtol=1e-12, atol=1e-12)
    pet_alt = o_pm.run_pmpet(d, n, int(g['nlcs']), int(g['start_year']), int(g['end_year']), 2, 7, lcy)
    assert np.allclose(pet_alt, g['pet_alt'], rtol=1e-12, atol=1e-12)


def test_pm_land_cover_year_and_leap():
    assert o_months.pm_land_cover_index(1994, [1990, 2000, 2005]) == 0
    assert o_months.pm_land_cover_index(1995, [1990, 2000, 2005]) == 1
    assert o_months.pm_land_cover_index(2005, [1990, 2000, 2005]) == 2
    assert o_months.pm_land_cover_index(2100, [2005, 1990, 2000]) == 2
    assert o_months.pm_days_in_month(1900)[1] == 28 and o_months.pm_days_in_month(2000)[1] == 29
    tab = o_months.set_month_arrays(24, 1900, 1901)
    assert tab[1, 2] == 29 and tab[13, 2] == 28          # the routing table's year % 4 rule (general.py:37)


@pytest.mark.parametrize('tag', ['snow', 'nosnow'])
def test_abcd_matches_reference(golden, tag):
    g = golden('abcd')
    tmin = g['tmin'] if tag == 'snow' else None
    pet, aet, q, sav = o_abcd.abcd_execute(int(g['n_basins']), g['basin_ids'], g['pet'], g['precip'], tmin,
                                           g['pars'], int(g['n_months']), int(g['spinup']), jobs=-1)
    assert np.array_equal(pet, g['pet'])
    for name, arr in (('aet', aet), ('q', q), ('sav', sav)):
        assert rel_err(arr, g[name + '_' + tag]) < 1e-12, name


def test_abcd_spinup_state_and_short_spinup(golden):
    g = golden('abcd')
    bid = g['basin_ids']
    he = o_abcd.ABCD(g['pars'][bid - 1], g['pet'], g['precip'], g['tmin'], bid, int(g['n_months']), int(g['spinup']))
    he.emulate()
    assert rel_err(he.sm0, g['sm0']) < 1e-13 and rel_err(he.gw0, g['gw0']) < 1e-13
    with pytest.raises(IndexError):
        o_abcd.ABCD(g['pars'][bid - 1], g['pet'], g['precip'], g['tmin'], bid, 60, 24).emulate()


@pytest.mark.parametrize('tag', ['rand', 'tree'])
def test_topology_matches_reference(golden, tag):
    g = golden('topo')
    st = SimpleNamespace(ngridrow=int(g['nrow']), ngridcol=int(g['ncol']))
    ds = o_mrtm.downstream(g[tag + '_coords'], g[tag + '_flowdir'], st)
    assert np.array_equal(ds, g[tag + '_dsid'])
    up = o_mrtm.upstream(g[tag + '_coords'], ds, st)
    assert np.array_equal(up, g[tag + '_upid'])
    um = o_mrtm.upstream_genmatrix(up).tocsr()
    um.sort_indices()
    assert np.array_equal(um.indptr, g[tag + '_um_indptr'])
    assert np.array_equal(um.indices, g[tag + '_um_indices'])
    assert np.array_equal(um.data, g[tag + '_um_data'])


def _um(g, t, tag):
    import scipy.sparse as sparse
    n = len(t[tag + '_dsid'])
    return sparse.csr_matrix((t[tag + '_um_data'], t[tag + '_um_indices'], t[tag + '_um_indptr']), shape=(n, n))


@pytest.mark.parametrize('tag', ['rand', 'tree'])
def test_streamrouting_bit_exact(golden, tag):
    g, t = golden('mrtm'), golden('topo')
    um = _um(g, t, tag)
    S = g[tag + '_S0']
    n = len(S)
    for nday in (28, 29, 30, 31):
        S, favg, F = o_mrtm.streamrouting(g[tag + '_L'], S, np.zeros(n), g[tag + '_chv'], g['%s_q_%d' % (tag, nday)],
                                          g[tag + '_area'], nday, 10800, um)
        assert np.array_equal(S, g['%s_S_%d' % (tag, nday)])
        assert np.array_equal(favg, g['%s_Favg_%d' % (tag, nday)])
        assert np.array_equal(F, g['%s_F_%d' % (tag, nday)])


@pytest.mark.parametrize('tag', ['rand', 'tree'])
def test_route_series_bit_exact(golden, tag):
    g, t = golden('mrtm'), golden('topo')
    chs, avg, fend = o_mrtm.route_series(_um(g, t, tag), g[tag + '_L'], g[tag + '_chv'], g[tag + '_area'],
                                         g[tag + '_series_runoff'], g[tag + '_series_ndays'],
                                         int(g['series_spinup']))
    assert np.array_equal(chs, g[tag + '_series_chstorage'])
    assert np.array_equal(avg, g[tag + '_series_avgchflow'])
    assert np.array_equal(fend, g[tag + '_series_Fend'])
    ndays = o_months.set_month_arrays(12, int(g['series_year']), int(g['series_year']))[:5, 2]
    assert np.array_equal(ndays, g[tag + '_series_ndays'])


@pytest.mark.parametrize('tag', ['rand', 'tree'])
def test_fused_routing_form_matches_reference_within_1e9(golden, tag):
    """The arithmetic of the reassociated kernel form (oracle.mrtm.streamrouting_fused: inflow as a plain sum, the update of
    mrtm.py:50-69 fused to eight operations) against the REFERENCE's own streamrouting outputs (tests/golden/mrtm.npz, 28-31
    day months, firing cells included): identical NaN masks, every value within 1e-9 |ref| (+ 1e-3 m3 / 1e-9 m3/s) -- the bar
    the device kernel of that form is held to on the GPU."""
    g, t = golden('mrtm'), golden('topo')
    um = _um(g, t, tag)
    S = g[tag + '_S0']
    n = len(S)
    worst = 0.0
    for nday in (28, 29, 30, 31):
        S1, favg, F = o_mrtm.streamrouting_fused(g[tag + '_L'], S, np.zeros(n), g[tag + '_chv'], g['%s_q_%d' % (tag, nday)],
                                                 g[tag + '_area'], nday, 10800, um)
        for x, name, atol in ((S1, 'S', 1e-3), (favg, 'Favg', 1e-9), (F, 'F', 1e-9)):
            ref = g['%s_%s_%d' % (tag, name, nday)]
            assert np.array_equal(np.isnan(x), np.isnan(ref))
            m = ~np.isnan(ref)
            err = np.abs(x[m] - ref[m])
            assert (err <= 1e-9 * np.abs(ref[m]) + atol).all(), (tag, nday, name, float(err.max()))
            big = np.abs(ref[m]) > 1e6 * atol
            if big.any():
                worst = max(worst, float((err[big] / np.abs(ref[m][big])).max()))
        S = g['%s_S_%d' % (tag, nday)]
    assert worst < 1e-11, worst


@pytest.mark.parametrize('basin', [0, 1])
@pytest.mark.parametrize('unit', ['km3_per_mth', 'mm_per_mth'])
@pytest.mark.parametrize('tag', ['snow', 'nosnow'])
def test_kge_objective_matches_reference(golden, basin, unit, tag):
    g = golden('kge')
    b = str(basin)
    tmin = g['tmin_' + b] if tag == 'snow' else None
    npar = 5 if tag == 'snow' else 4
    for k, p in enumerate(g['pars_' + b]):
        series = o_calib.basin_runoff(p[:npar], 0, g['pet_' + b], g['precip_' + b], tmin, int(g['n_months']),
                                      int(g['spinup']), unit, g['areas_' + b])
        assert np.allclose(series, g['series_%s_%s_%s' % (b, unit, tag)][k], rtol=1e-12, atol=0)
        ed = o_calib.kge_distance(series, g['robs_' + b])
        assert abs(ed - g['ed_%s_%s_%s' % (b, unit, tag)][k]) < 1e-12


def test_writer_aggregation_matches_reference(golden):
    from oracle import writer as o_writer
    g = golden('writer')
    assert np.allclose(o_writer.agg_to_year(g['q'], 'sum'), g['ysum'], rtol=1e-13, atol=0)
    assert np.allclose(o_writer.agg_to_year(g['q'], 'mean'), g['ymean'], rtol=1e-13, atol=0, equal_nan=True)
    assert np.allclose(o_writer.mm_to_km3(g['q'], g['area']), g['km3'], rtol=1e-15, atol=0, equal_nan=True)
    sp = o_writer.agg_spatial(o_writer.mm_to_km3(o_writer.agg_to_year(g['q'], 'sum'), g['area']), g['ids'], 8)
    assert np.allclose(sp, g['spatial'], rtol=1e-13, atol=0, equal_nan=True)
    assert np.isnan(sp[4]).all() and np.isnan(sp[7]).all()          # ids 5 and 8 have no cells


def test_drought_matches_reference(golden):
    """oracle/drought.py against DroughtStats.getthresh / calculate_thresholds / droughtstats of the reference."""
    from oracle import drought as o_dr
    g = golden('drought')
    h = g['hydro']
    y0 = int(g['start_year'])
    assert np.array_equal(o_dr.calculate_thresholds(h, y0, 1975, 1994, 12), g['thresh12'], equal_nan=True)
    assert np.array_equal(o_dr.calculate_thresholds(h, y0, 1971, 1990, 1), g['thresh1'], equal_nan=True)
    assert np.array_equal(o_dr.getthresh(h[:240], 12, quantile=0.25), g['thresh12_q25'], equal_nan=True)
    assert np.array_equal(o_dr.getthresh(h[:240], 4, quantile=0.5), g['thresh4_q50'], equal_nan=True)
    for tag in ('thresh12', 'thresh1'):
        S, I, D = o_dr.droughtstats(h, g[tag])
        assert np.array_equal(S, g[tag + '_S']) and np.array_equal(I, g[tag + '_I']) and np.array_equal(D, g[tag + '_D'])
    assert g['thresh12_D'].max() >= 5 and (g['thresh12_D'][:, 3] == 0).all()       # droughts occur; NaN cell has none


def test_accessible_water_matches_reference(golden):
    """oracle/accessible.py against the csv the reference's AccessibleWater wrote, and its moving mean."""
    from oracle import accessible as o_ac
    g = golden('accessible')
    y0, y1, hist, g0, g1, step, window = (int(v) for v in g['settings'])
    table, totals = o_ac.accessible_water(g['runoff'], g['area'], g['ids'], g['bfi'], g['res'].reshape(-1, 1), y0, y1, hist,
                                          g0, g1, step, window, float(g['env_pct']))
    assert table.shape == g['table'].shape
    assert np.array_equal(table, g['table'])            # the csv holds shortest round-trip decimals (ndarray.astype(str))
    assert (totals[3] == 0).all() and table.max() > 0   # basin 4 has no cells
    assert np.array_equal(o_ac.rolling_mean_rows(g['demo'], 5), g['demo_roll5'])
    assert np.array_equal(o_ac.rolling_mean_rows(g['demo'], 9), g['demo_roll9'])


def test_route_series_by_network_equals_serial():
    """The full-size checker routes river networks in worker processes (oracle.mrtm.route_series_by_network: the bench's
    CPU child and the full-grid GPU tests): every output bit equals the serial month loops of route_series -- NaN runoff,
    cells that fire (velocity * dt / length > 1), spin-up and initial storage included."""
    from types import SimpleNamespace as NS
    from oracle import months, mrtm as o
    from xanthos_amd import synth
    w = synth.make_world(nrow=60, ncol=120, ncell=3000, n_basins=5, seed=3, outlet_frac=0.02)
    st = NS(ngridrow=w.nrow, ngridcol=w.ncol)
    um = o.upstream_genmatrix(o.upstream(w.coords, o.downstream(w.coords, w.flow_dir, st), st)).tocsr()
    rng = np.random.default_rng(1)
    q = rng.gamma(2.0, 30.0, (w.ncell, 12))
    q[rng.random(w.ncell) < 0.01] = np.nan
    s0 = rng.gamma(2.0, 1e5, w.ncell)
    nd = months.set_month_arrays(12, 1973, 1973)[:, 2]
    assert (w.velocity * 10800.0 / w.flow_dist > 1.0).sum() > 5           # cells that fire are in the world
    for S0, procs in ((None, 1), (None, 3), (s0, 4)):
        a = o.route_series(um, w.flow_dist, w.velocity, w.area, q, nd, 2, S0)
        b = o.route_series_by_network(um, w.flow_dist, w.velocity, w.area, q, nd, 2, S0, n_procs=procs)
        assert b[3] > 0
        for x, y in zip(a, b[:3]):
            assert np.array_equal(x, y, equal_nan=True)
    groups = o.network_groups(um, 4)
    assert sorted(np.concatenate(groups).tolist()) == list(range(w.ncell))
