"""GPU end-to-end (-m gpu): run_model() on a synthetic pm_abcd_mrtm input tree against the oracle chain, the
calibration driver, and full-size (67,420-cell) checks of each stage.

Everything here runs what the library ships -- since round 5 the reassociated routing form, since round 6 on prepared plans
(single running sums, folded leaves) -- and holds the routed values to that form's bar: identical NaN masks,
|x - ref| <= 1e-9 |ref| + 1e-3 m3 (storage) / 1e-9 m3/s (flow).  The bit-exact kernels are tested by flag elsewhere."""
from types import SimpleNamespace

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def routed_close(x, ref, atol, rtol=1e-9, tag=''):
    x, ref = np.asarray(x), np.asarray(ref)
    assert x.shape == ref.shape
    assert np.array_equal(np.isnan(x), np.isnan(ref)), 'NaN pattern differs ' + str(tag)
    m = ~np.isnan(ref)
    excess = np.abs(x[m] - ref[m]) - (atol + rtol * np.abs(ref[m]))
    assert (excess <= 0).all(), '{}: {} values beyond the bar, largest excess {:.3e}'.format(tag, int((excess > 0).sum()), excess.max())


def rel(x, ref):
    assert np.array_equal(np.isnan(x), np.isnan(ref))
    m = ~np.isnan(ref)
    return float(np.max(np.abs(x[m] - ref[m]) / (np.abs(ref[m]) + 1e-9))) if m.any() else 0.0


@pytest.fixture(scope='module')
def example(tmp_path_factory):
    from xanthos_amd import synth
    root = str(tmp_path_factory.mktemp('xanthos_example'))
    w = synth.make_world(nrow=36, ncol=72, ncell=900, n_basins=7, seed=33)
    f = synth.make_forcing(w, 36)
    return root, w, f, synth.write_example(root, w, f, 1971, 1973, runoff_spinup=25, routing_spinup=6)


def test_run_model_matches_oracle_chain(example):
    from oracle import abcd as o_abcd, months as o_months, mrtm as o_mrtm, pm as o_pm
    from xanthos_amd import Xanthos, synth
    root, w, f, ini = example
    res = Xanthos(ini).execute()
    assert res.Q.shape == (900, 36) and res.Avg_ChFlow.shape == (900, 36)
    d = synth.data_bag(w, f)
    pet = o_pm.run_pmpet(d, w.ncell, w.nlcs, 1971, 1973, 0, 6, w.lc_years)
    _, aet, q, sav = o_abcd.abcd_execute(w.n_basins, w.basin_ids, pet, f['precip'], f['abcd_tmin'], w.abcd_pars, 36, 25, -1)
    assert rel(res.PET, pet) < 1e-10 and rel(res.AET, aet) < 1e-8 and rel(res.Q, q) < 1e-8 and rel(res.Sav, sav) < 1e-8
    # routing for identical runoff (the default form: within its bar): feed the GPU's own Q to the oracle
    st = SimpleNamespace(ngridrow=w.nrow, ngridcol=w.ncol)
    um = o_mrtm.upstream_genmatrix(o_mrtm.upstream(w.coords, o_mrtm.downstream(w.coords, w.flow_dir, st), st))
    ndays = o_months.set_month_arrays(36, 1971, 1973)[:, 2]
    # (the loader's ha -> km2 conversion rounds the areas by an ulp, so take the arrays the run actually used)
    chs, avg, _ = o_mrtm.route_series(um, res.data.flow_dist, res.data.str_velocity, res.data.area, res.Q, ndays, 6)
    routed_close(res.ChStorage, chs, 1e-3, tag='ChStorage')
    routed_close(res.Avg_ChFlow, avg, 1e-9, tag='Avg_ChFlow')
    info = res.pipe.plan.info()
    assert info['last_tree_kernel'] == 4 and res.pipe.plan.rsum_info()['pair_cells'] >= 0, (info, res.pipe.plan.rsum_info())
    # per-stage plugin calls give the same arrays as the device-resident simulation
    from xanthos_amd.components import Components
    from xanthos_amd.ini_reader import ConfigReader
    c = Components(ConfigReader(ini))
    pet2 = c.calculate_pet()
    c.calculate_runoff(pet=pet2)
    assert np.array_equal(pet2, res.PET) and np.array_equal(c.Q, res.Q, equal_nan=True)
    routed_close(c.calculate_routing(c.Q), res.Avg_ChFlow, 1e-9, tag='plugin calls')
    import os
    out_csv = os.path.join(root, 'output', 'pm_abcd_mrtm_synth', 'q_mmpermonth_pm_abcd_mrtm_synth.csv')      # reference naming
    assert os.path.isfile(out_csv)
    first = open(out_csv).read().splitlines()[:2]
    assert first[0].startswith('id,197101,197102') and first[1].startswith('1,')


def test_in_memory_forcing_override(example):
    """execute(args) replaces file settings by arrays, like the reference's test entry (model.py:82-98)."""
    from xanthos_amd import Xanthos
    root, w, f, ini = example
    base = Xanthos(ini).execute()
    wet = Xanthos(ini).execute({'PrecipitationFile': f['precip'] * 2.0})
    m = ~np.isnan(base.Q)
    assert (wet.Q[m] >= base.Q[m] - 1e-9).all() and wet.Q[m].sum() > 1.5 * base.Q[m].sum()
    # missing values in the forcing: nan_to_num happens on the device and must equal the loader's host transform
    holes = {k: f[k].copy() for k in ('tas', 'tmin', 'rhs', 'abcd_tmin')}
    for k, a in holes.items():
        a[::7, ::5] = np.nan
    a = Xanthos(ini).execute({'pm_tas': holes['tas'], 'pm_tmin': holes['tmin'], 'pm_rhs': holes['rhs'],
                              'TempMinFile': holes['abcd_tmin']})
    b = Xanthos(ini).execute({'pm_tas': np.nan_to_num(holes['tas']), 'pm_tmin': np.nan_to_num(holes['tmin']),
                              'pm_rhs': np.nan_to_num(holes['rhs']), 'TempMinFile': np.nan_to_num(holes['abcd_tmin']),
                              'device_transforms': False})
    for name in ('PET', 'Q', 'Avg_ChFlow'):
        assert np.array_equal(getattr(a, name), getattr(b, name), equal_nan=True), name
    assert not np.isnan(a.PET).any()


def test_calibration_recovers_kge(example):
    """DE over the batched GPU objective reaches KGE ~ 1 on observations generated from known parameters."""
    from oracle import calib as o_calib
    from xanthos_amd import synth
    from xanthos_amd.calibrate.calibrate_abcd import Calibrate, objective_kge
    root, w, f, ini = example
    nm, spin, basin = 36, 25, int(np.argmax(np.bincount(w.basin_ids)[1:]) + 1)
    sel = w.basin_ids == basin
    ok = sel & ~np.isnan(f['precip']).any(axis=1)
    pet = np.random.default_rng(0).uniform(20, 150, (w.ncell, nm))
    truth = np.array([0.97, 0.6, 0.4, 0.3, 0.5])
    series = o_calib.basin_runoff(truth, 0, pet[sel], f['precip'][sel], f['abcd_tmin'][sel], nm, spin, 'km3_per_mth',
                                  w.area[sel])
    obs = np.stack([np.full(nm, basin), np.zeros(nm), np.zeros(nm), series], axis=1)
    ed = objective_kge(truth, pet[sel], f['precip'][sel], f['abcd_tmin'][sel], nm, spin, 'km3_per_mth', w.area[sel], series)
    assert ed < 1e-9 and ok.any()
    cal = Calibrate(basin, w.basin_ids, w.area, f['precip'], pet, obs[:, [0, 3]], f['abcd_tmin'], nm, spin, 0,
                    'km3_per_mth', None, seed=7)
    cal.calibrate_basin(popsize=15)
    assert cal.kge_vals[0] > 0.99, cal.kge_vals
    assert cal.nfev >= 75


def test_full_size_grid_each_stage():
    """67,420 cells: PM and ABCD on sampled cells / basins vs the oracle, routing 8 months vs scipy (the default form, through
    the plugin API: routing.mrtm.route_series prepares its plan from the L, ChV and dt it holds)."""
    from oracle import abcd as o_abcd, mrtm as o_mrtm, pm as o_pm
    from xanthos_amd import _hip, synth
    from xanthos_amd.pipeline import pipeline_from_world
    ctx = _hip.get_context(0)
    w = synth.make_world()
    nm = 48
    pipe = pipeline_from_world(ctx, w, nm, 1961, 30, 4)
    fdev = pipe.alloc_forcing()
    ctx.synth_forcing(11, pipe.ncell, nm, ctx.upload(w.latitude), fdev, nan_frac=0.001)
    pipe.run()
    info = pipe.plan.info()
    assert info['flow_cells'] == w.ncell and info['fallback_cells'] == 0
    cells = np.arange(5000, 5600)                                   # contiguous: tairprev = previous cell
    fh = {k: pipe.rows(fdev[k], np.arange(4999, 5600)) for k in ('tas', 'tmin', 'rhs', 'wind', 'rsds', 'rlds')}
    sub = SimpleNamespace(**{k: getattr(w, k) for k in ('cL', 'beta', 'rslimit', 'ae', 'be', 'Tminopen', 'Tminclose',
                                                         'VPDclose', 'VPDopen', 'RBLmin', 'RBLmax', 'rc', 'emiss',
                                                         'alpha', 'lai', 'laimax', 'laimin')})
    sub.elev, sub.lct = w.elev[4999:5600], w.lct[4999:5600]
    d = synth.data_bag(sub, fh)
    ref = o_pm.run_pmpet(d, 601, w.nlcs, 1961, 1964, 0, 6, w.lc_years)[1:]
    assert rel(pipe.rows(pipe.out['pet'], cells), ref) < 1e-10
    basins = [3, 77, 150]
    bc = np.nonzero(np.isin(w.basin_ids, basins))[0]
    remap = {b: i + 1 for i, b in enumerate(basins)}
    aet, q, sav = o_abcd.abcd_parallel(3, w.abcd_pars[np.array(basins) - 1], np.array([remap[b] for b in w.basin_ids[bc]]),
                                       pipe.rows(pipe.out['pet'], bc), pipe.rows(fdev['precip'], bc),
                                       pipe.rows(fdev['abcd_tmin'], bc), nm, 30, jobs=1)
    assert rel(pipe.rows(pipe.out['q'], bc), q) < 1e-8 and rel(pipe.rows(pipe.out['sav'], bc), sav) < 1e-8
    qh = pipe.out['q'].download()
    chs, avg, _ = o_mrtm.route_series(pipe.um.tocsr(), w.flow_dist, w.velocity, w.area, qh[:, :8].copy(),
                                      pipe.ndays[:8], 0)
    from xanthos_amd.routing import mrtm
    g_chs, g_avg, _ = mrtm.route_series(pipe.um, w.flow_dist, w.velocity, w.area, qh[:, :8].copy(), pipe.ndays[:8], 0)
    routed_close(g_chs, chs, 1e-3, tag='chs')
    routed_close(g_avg, avg, 1e-9, tag='avg')
    ri = pipe.plan.rsum_info()
    assert pipe.plan.info()['last_tree_kernel'] == 4 and ri['folded'] > 0 and ri['pair_cells'] >= 0 and ri['fold_disabled'] == 0, ri
    # size-independent property: routing conserves water -- storage change = inflow - outflow at the outlets
    assert np.isnan(g_avg).sum() == np.isnan(avg).sum()


def test_basin_sharded_run_equals_whole_world():
    """Shard one world into 3 basin/network-closed shards, run each shard's pipeline on this GPU, reassemble:
    PET / AET / Q / Sav are bit-identical to the unsharded run (what rank 0 holds after the gather on a multi-GPU node), the
    routed outputs within the default form's bar (its sums follow the partition, and a shard's partition is not the world's;
    with XH_ROUTE_EXACT they are bit-identical too: test_gpu_multirank.py)."""
    from xanthos_amd import _hip, synth
    from xanthos_amd.dist import fill_shard_forcing, make_shards, sub_world
    from xanthos_amd.pipeline import OUTPUTS, pipeline_from_world, topology_from_world
    ctx = _hip.get_context(0)
    w = synth.make_world(nrow=60, ncol=120, ncell=4000, n_basins=11, seed=8)
    um = topology_from_world(w)
    nm, seed = 36, 99
    whole = pipeline_from_world(ctx, w, nm, 1971, 25, 6, um=um)
    ctx.synth_forcing(seed, w.ncell, nm, ctx.upload(w.latitude), whole.alloc_forcing(), nan_frac=0.002)
    whole.run()
    ref = whole.download()
    got = {k: np.full((w.ncell, nm), -1.0) for k in OUTPUTS}
    shards = make_shards(w, um, 3)
    assert min(len(s.cells) for s in shards) > 0
    for s in shards:
        sw, sum_ = sub_world(w, um, s)
        pipe = pipeline_from_world(ctx, sw, nm, 1971, 25, 6, um=sum_)
        pipe.alloc_forcing()
        fill_shard_forcing(ctx, w, s, pipe, seed, nan_frac=0.002)
        pipe.run()
        out = pipe.download()
        for k in OUTPUTS:
            got[k][s.cells] = out[k]
    for k in OUTPUTS:
        if k in ('chs', 'avg'):
            routed_close(got[k], ref[k], 1e-3 if k == 'chs' else 1e-9, tag=k)
        else:
            assert np.array_equal(got[k], ref[k], equal_nan=True), k


def test_calibrate_all_lockstep(example, tmp_path):
    """calibrate_all: every basin of the example searched in lock-step by the device-side DE; the files the
    reference writes per basin (calibrate_abcd.py:130-131) appear and KGE is high where data is clean."""
    from types import SimpleNamespace as NS
    from oracle import calib as o_calib
    from xanthos_amd.calibrate.calibrate_abcd import calibrate_all
    root, w, f, ini = example
    nm, spin = 36, 25
    pet = np.random.default_rng(1).uniform(20, 150, (w.ncell, nm))
    truth = np.array([0.96, 0.8, 0.5, 0.4, 0.3])
    rows = []
    for b in (1, 2, 3):
        sel = w.basin_ids == b
        series = o_calib.basin_runoff(truth, 0, pet[sel], f['precip'][sel], f['abcd_tmin'][sel], nm, spin, 'km3_per_mth',
                                      w.area[sel])
        rows.append(np.stack([np.full(nm, b), np.zeros(nm), np.zeros(nm), series], axis=1))
    obs = np.concatenate(rows)
    settings = NS(set_calibrate=0, obs_unit='km3_per_mth', cal_basins=['1-3'], nmonths=nm, runoff_spinup=spin,
                  calib_out_dir=str(tmp_path), device=0)
    data = NS(basin_ids=w.basin_ids, area=w.area, precip=f['precip'], tmin=f['abcd_tmin'], cal_obs=obs[:, [0, 3]])
    res = calibrate_all(settings, data, pet, seed=11)
    assert sorted(res) == [1, 2, 3]
    for b, (x, kge) in res.items():
        assert kge > 0.98, (b, kge)
        assert np.load(str(tmp_path / 'kge_result_basin_{}.npy'.format(b)))[0] == kge
        assert np.load(str(tmp_path / 'abcdm_parameters_basin_{}.npy'.format(b))).shape == (1, 5)


def test_run_model_with_post_processors(tmp_path):
    """CalculateDroughtStats / CalculateAccessibleWater = 1 in the ini: the files the reference would write, with the
    values the oracle gives for the run's own runoff (thresholds run, then the statistics run from its file)."""
    import os
    from oracle import accessible as o_ac, drought as o_dr
    from xanthos_amd import Xanthos, synth
    root = str(tmp_path)
    w = synth.make_world(nrow=36, ncol=72, ncell=900, n_basins=7, seed=34)
    f = synth.make_forcing(w, 72)
    ini = synth.write_example(root, w, f, 1971, 1976, runoff_spinup=25, routing_spinup=6, post=True)
    res = Xanthos(ini).execute()
    out = os.path.join(root, 'output', 'pm_abcd_mrtm_synth')
    th = np.load(os.path.join(out, 'drought_thresholds_pm_abcd_mrtm_synth.npy'))
    assert np.array_equal(th, o_dr.calculate_thresholds(res.Q.T, 1971, 1971, 1976, 12), equal_nan=True)
    import pandas as pd
    acc = os.path.join(root, 'input', 'accessible')
    cap = pd.read_csv(os.path.join(acc, 'total_reservoir_storage.csv'), header=None).values
    bfi = pd.read_csv(os.path.join(acc, 'bfi_per_basin.csv'))['bfi_avg'].values
    table, _ = o_ac.accessible_water(res.Q, res.data.area, res.data.basin_ids, bfi, cap, 1971, 1976, 1976, 1971, 1976, 1, 3, 0.1)
    lines = open(os.path.join(out, 'accessible_water_km3peryr_pm_abcd_mrtm_synth.csv')).read().splitlines()
    assert lines[0] == 'id,name,1971,1972,1973,1974,1975,1976' and lines[1].startswith('1,Basin 001,')
    got = np.array([[float(v) for v in ln.split(',')[2:]] for ln in lines[1:]])
    assert np.array_equal(got, table)
    # second run: statistics from the thresholds file of the first
    x = Xanthos(ini)
    stats = x.execute({'drought_thresholds': os.path.join(out, 'drought_thresholds_pm_abcd_mrtm_synth.npy'),
                       'CalculateAccessibleWater': 0, 'OutputFormat': 4})
    S, I, D = o_dr.droughtstats(stats.Q.T, th)
    for name, ref in (('severity', S), ('intensity', I), ('duration', D)):
        got = np.load(os.path.join(out, 'drought_{}_pm_abcd_mrtm_synth.npy'.format(name)))
        assert np.array_equal(got.T, ref)
    assert D.max() >= 1


def test_run_model_aggregates_and_future_mode(tmp_path):
    """AggregateRunoffBasin / Country / GCAMRegion = 1 (out_writer.py:126-158: one row per NAME, basins and regions from
    id 1, countries from id 0, a name without cells gives NaN) and HistFlag = False with a ChStorageFile: routing starts
    from the last column of the historical channel storage (data_load.py:427-438)."""
    import os
    from oracle import months as o_months, mrtm as o_mrtm
    from xanthos_amd import Xanthos, synth
    root = str(tmp_path)
    w = synth.make_world(nrow=36, ncol=72, ncell=900, n_basins=7, seed=35)
    f = synth.make_forcing(w, 36)
    chs0 = np.random.default_rng(2).uniform(0, 5e7, (w.ncell, 4))
    ini = synth.write_example(root, w, f, 1971, 1973, runoff_spinup=25, routing_spinup=6, aggregates=True,
                              hist_flag=False, ch_storage=chs0)
    res = Xanthos(ini).execute()
    assert np.array_equal(res.data.chs_prev, chs0[:, -1])
    st = SimpleNamespace(ngridrow=w.nrow, ngridcol=w.ncol)
    um = o_mrtm.upstream_genmatrix(o_mrtm.upstream(w.coords, o_mrtm.downstream(w.coords, w.flow_dir, st), st))
    ndays = o_months.set_month_arrays(36, 1971, 1973)[:, 2]
    chs, avg, _ = o_mrtm.route_series(um, res.data.flow_dist, res.data.str_velocity, res.data.area, res.Q, ndays, 6,
                                      S0=chs0[:, -1])
    routed_close(res.ChStorage, chs, 1e-3, tag='ChStorage')
    routed_close(res.Avg_ChFlow, avg, 1e-9, tag='Avg_ChFlow')
    zero = o_mrtm.route_series(um, res.data.flow_dist, res.data.str_velocity, res.data.area, res.Q, ndays, 6)[0]
    assert not np.array_equal(zero, chs, equal_nan=True)
    out = os.path.join(root, 'output', 'pm_abcd_mrtm_synth')
    for fname, ids, first, n in (('Basin_runoff', w.basin_ids, 1, 7), ('Country_runoff', w.basin_ids % 9, 0, 10),
                                 ('GCAMRegion_runoff', w.basin_ids % 6 + 1, 1, 7)):
        lines = open(os.path.join(out, fname + '_mmpermonth_pm_abcd_mrtm_synth.csv')).read().splitlines()
        assert lines[0].startswith('id,name,197101,197102') and len(lines) == n + 1
        for k, ln in enumerate(lines[1:]):
            cols = ln.split(',')
            assert int(cols[0]) == first + k
            sel = ids == first + k
            vals = np.array([float(v) if v != '' else np.nan for v in cols[2:]])
            if sel.any():
                want = np.nansum(res.Q[sel], axis=0)
                assert np.allclose(vals, want, rtol=1e-12, atol=1e-12), (fname, k)
            else:
                assert np.isnan(vals).all(), (fname, k)                 # a name without cells
    assert open(os.path.join(out, 'Country_runoff_mmpermonth_pm_abcd_mrtm_synth.csv')).read().splitlines()[1] \
        .startswith('0,Country 0,')


@pytest.mark.parametrize('nm,block', [(120, 48), (36, 48), (96, 96)])
def test_fused_pipeline_equals_stage_by_stage(nm, block):
    """xh_run_fused (PM blocks, the ABCD march one block behind with its state carried across blocks, routing started on
    the first block of runoff and polling for the months behind it) gives bit for bit what xh_pm_pet + xh_abcd +
    xh_route_series give, for series of several blocks, of less than one block, and of exactly one."""
    from xanthos_amd import _hip, synth
    from xanthos_amd.pipeline import OUTPUTS, pipeline_from_world
    ctx = _hip.get_context(0)
    w = synth.make_world(nrow=60, ncol=120, ncell=4000, n_basins=11, seed=8)
    pipe = pipeline_from_world(ctx, w, nm, 1971, 25, 6)
    ctx.synth_forcing(17, w.ncell, nm, ctx.upload(w.latitude), pipe.alloc_forcing(), nan_frac=0.002)
    pipe.run(fused=False)
    ref = pipe.download()
    for rep in range(3):
        for k in OUTPUTS:
            pipe.out[k].zero()
        pipe.run_fused(block_months=block)
        got = pipe.download()
        for k in OUTPUTS:
            assert np.array_equal(got[k], ref[k], equal_nan=True), (k, rep)
    assert pipe.plan.info()['reroutes'] == 0
    # PM + ABCD only
    for k in OUTPUTS:
        pipe.out[k].zero()
    pipe.run(('pm', 'abcd'))
    got = pipe.download(('pet', 'aet', 'q', 'sav'))
    for k in ('pet', 'aet', 'q', 'sav'):
        assert np.array_equal(got[k], ref[k], equal_nan=True), k


def test_file_movers_round_trip(tmp_path):
    """xh_upload_file / xh_download_file: the body of a .npy to HBM and back, several 8 MiB chunks with a ragged tail,
    byte-exact; a short or missing file is an error, not a partial array (loader: data_load.py:186-195, :342-350)."""
    from xanthos_amd import _hip
    ctx = _hip.get_context(0)
    rng = np.random.default_rng(5)
    for shape in ((3, 5), (1201, 2503)):                      # 120 B and 24 MB (3 chunks, the last one 7.2 MiB short)
        a = rng.standard_normal(shape)
        a[0, 0] = np.nan
        src = str(tmp_path / 'in_{}.npy'.format(shape[0]))
        np.save(src, a)
        mm = np.load(src, mmap_mode='r')
        d = ctx.empty(shape)
        ctx.upload_file(d, mm.filename, mm.offset, mm.nbytes, threads=3)
        assert np.array_equal(d.download(), a, equal_nan=True)
        dst = str(tmp_path / 'out_{}.npy'.format(shape[0]))
        ctx.save_npy(dst, d)
        assert np.array_equal(np.load(dst), a, equal_nan=True)
        d2 = ctx.upload(a[::-1].copy())                        # two files side by side (xh_download_files)
        pair = [str(tmp_path / 'p0_{}.npy'.format(shape[0])), str(tmp_path / 'p1_{}.npy'.format(shape[0]))]
        ctx.save_npy_many([(pair[0], d), (pair[1], d2)])
        assert np.array_equal(np.load(pair[0]), a, equal_nan=True) and np.array_equal(np.load(pair[1]), a[::-1], equal_nan=True)
        d2.free()
        with pytest.raises(RuntimeError):
            ctx.upload_file(d, mm.filename, mm.offset + 8, mm.nbytes)           # runs past the end of the file
        with pytest.raises(ValueError):
            ctx.upload_file(d, mm.filename, mm.offset, mm.nbytes - 8)
        d.free()
    d = ctx.empty((4,))
    with pytest.raises(RuntimeError):
        ctx.upload_file(d, str(tmp_path / 'missing.npy'), 0, 32)
    d.free()
