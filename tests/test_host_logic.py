"""CPU tests of the host-side mirror: .ini surface, selector validation, month tables, DE driver, sharding math."""
import os
import re

import numpy as np
import pytest

from xanthos_amd import synth
from xanthos_amd.calibrate.calibrate_abcd import assign_basins, expand_str_range
from xanthos_amd.ini_reader import ConfigReader, ValidationException, parse_ini

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))


@pytest.fixture(scope='module')
def example(tmp_path_factory):
    root = str(tmp_path_factory.mktemp('xanthos_example'))
    w = synth.make_world(nrow=24, ncol=48, ncell=300, n_basins=5, seed=21)
    f = synth.make_forcing(w, 36)
    ini = synth.write_example(root, w, f, 1971, 1973, runoff_spinup=25, routing_spinup=6)
    return root, w, f, ini


def test_ini_surface(example):
    root, w, f, ini = example
    raw = parse_ini(ini)
    assert raw['PET']['penman-monteith']['pm_lc_years'] == ['1970', '1990', '2005']
    c = ConfigReader(ini)
    assert (c.pet_module, c.runoff_module, c.routing_module, c.mod_cfg) == ('pm', 'abcd', 'mrtm', 'pm_abcd_mrtm')
    assert c.nmonths == 36 and c.ncell == 300 and c.n_basins == 5
    assert c.pm_nlcs == 8 and c.pm_water_idx == 0 and c.pm_snow_idx == 6 and c.pm_lc_years == [1970, 1990, 2005]
    assert c.runoff_spinup == 25 and c.routing_spinup == 6 and c.ro_jobs == -1
    assert c.pm_params.endswith(os.path.join('penman_monteith', 'gcam_ET_para.csv'))
    assert c.calib_file.endswith(os.path.join('abcd', 'pars.npy'))
    assert c.output_vars == ['q', 'avgchflow'] and c.calibrate == 0
    c.update({'StartYear': 1972})
    assert c.StartYear == 1972


def test_ini_defaults_and_selector_validation(example, tmp_path):
    root, w, f, ini = example
    text = open(ini).read()

    def variant(old, new):
        p = tmp_path / 'v.ini'
        p.write_text(text.replace(old, new))
        return str(p)
    assert ConfigReader(variant('routing_spinup = 6\n', '')).routing_spinup == 36       # default = nmonths (:412-415)
    assert ConfigReader(variant('pet_module = pm', 'pet_module = PM')).pet_module == 'pm'   # lower-cased (:214)
    for bad in ('pet_module = hargreaves', 'pet_module = nosuch'):
        with pytest.raises(ValidationException):
            ConfigReader(variant('pet_module = pm', bad))
    with pytest.raises(ValidationException):
        ConfigReader(variant('runoff_module = abcd', 'runoff_module = gwam'))
    with pytest.raises(ValidationException):
        ConfigReader(variant('routing_module = mrtm', 'routing_module = rtm'))
    # round 5: which form of the routing kernel ([[mrtm]] routing_form; not a key of the reference)
    assert ConfigReader(ini).routing_form == 'default'
    for value, flag in (('reassociated', 128), ('Exact', 256), ('default', 0)):
        cfg = ConfigReader(variant('routing_spinup = 6', 'routing_form = {}\nrouting_spinup = 6'.format(value)))
        assert cfg.routing_form == value.lower()
        from xanthos_amd.components import Components
        from types import SimpleNamespace as NS
        assert Components.route_flags(NS(s=cfg)) == flag
    with pytest.raises(ValidationException):
        ConfigReader(variant('routing_spinup = 6', 'routing_form = fast\nrouting_spinup = 6'))


def test_data_loader_transforms(example):
    from xanthos_amd.data_load import DataLoader
    root, w, f, ini = example
    d = DataLoader(ConfigReader(ini))
    assert np.allclose(d.area, w.area) and np.array_equal(d.basin_ids, w.basin_ids)
    assert np.array_equal(d.tairprev_load[1:], d.tair_load[:-1]) and not d.tairprev_load[0].any()
    assert d.flow_dist.min() >= 1000 and d.str_velocity.min() >= 0
    assert np.isnan(d.precip).sum() == np.isnan(f['precip']).sum()            # precipitation keeps NaN
    assert d.elev.shape == (300, 1) and d.lct_load.shape == (300, 8, 3)


def test_calibration_basin_assignment():
    """Basins are dealt to the ranks largest-first onto the least-loaded rank (calibration fan-out)."""
    assert expand_str_range(['0-2', '6', '7-9']) == [0, 1, 2, 6, 7, 8, 9]
    sizes = np.array([5, 900, 20, 310, 300, 40, 290, 7])
    owner = assign_basins(sizes, 3)
    load = np.bincount(owner, weights=sizes, minlength=3)
    assert owner[1] != owner[3] and load.max() <= sizes.sum() / 3 + sizes.max()       # LPT bound
    assert set(owner) == {0, 1, 2} and np.array_equal(assign_basins(sizes, 1), np.zeros(8, dtype=int))
    # deterministic: every rank computes the same table without talking to the others
    assert np.array_equal(owner, assign_basins(sizes.copy(), 3))


def test_netcdf_and_mat_inputs(tmp_path):
    """PrecipitationFile / TempMinFile may be NetCDF-classic or MATLAB files with a variable name (data_load.py:366-384)."""
    import scipy.io as sio
    from xanthos_amd.data_load import load_file
    a = np.arange(12.0).reshape(3, 4)
    nc = str(tmp_path / 'pr.nc')
    f = sio.netcdf_file(nc, 'w')
    f.createDimension('c', 3)
    f.createDimension('m', 4)
    v = f.createVariable('pr', 'd', ('c', 'm'))
    v[:] = a
    f.close()
    got = load_file(nc, key='pr')
    assert np.array_equal(got, a) and got.dtype.byteorder in ('=', '<', '|')
    mat = str(tmp_path / 'tmin.mat')
    sio.savemat(mat, {'tmin': a})
    assert np.array_equal(load_file(mat, key='tmin'), a)


def test_post_processor_ini_sections(tmp_path):
    """[Drought] / [AccessibleWater] parsing and validation (ini_reader.py:460-486, :547-551); other post-processors
    are rejected by name."""
    from xanthos_amd import synth
    from xanthos_amd.drought.drought_stats import quantile_plan
    from xanthos_amd.ini_reader import ConfigReader, ValidationException, parse_ini
    w = synth.make_world(nrow=24, ncol=48, ncell=300, n_basins=4, seed=2)
    f = synth.make_forcing(w, 36)
    ini = synth.write_example(str(tmp_path), w, f, 1971, 1973, runoff_spinup=25, routing_spinup=6, post=True)
    c = ConfigReader(ini)
    assert (c.CalculateDroughtStats, c.CalculateAccessibleWater) == (1, 1)
    assert c.drought_var == 'q' and c.drought_thresholds is None and c.threshold_nper == 12
    assert (c.threshold_start_year, c.threshold_end_year) == (1971, 1973)
    assert c.ResCapacityFile.endswith('input/accessible/total_reservoir_storage.csv')
    assert (c.GCAM_StartYear, c.GCAM_EndYear, c.GCAM_YearStep, c.MovingMeanWindow, c.Env_FlowPercent) == (1971, 1973, 1, 3, 0.1)
    d = parse_ini(ini)
    d['Drought']['threshold_end_year'] = '1990'
    with pytest.raises(ValidationException, match='Drought threshold year range'):
        ConfigReader(d)
    d = parse_ini(ini)
    d['AccessibleWater']['GCAM_StartYear'] = '1960'
    with pytest.raises(ValidationException, match='outside the range of years'):
        ConfigReader(d)
    d = parse_ini(ini)
    d['Project']['PerformDiagnostics'] = '1'
    with pytest.raises(ValidationException, match='PerformDiagnostics'):
        ConfigReader(d)
    # numpy's linear quantile: (k_prev, k_next, weight) for the sample sizes the thresholds use
    import numpy as np
    for n, q in ((20, 0.1), (30, 0.1), (20, 0.25), (60, 0.5), (5, 0.1), (1, 0.1), (11, 1.0)):
        k0, k1, g = quantile_plan(n, (q * 100) / 100.0)
        x = np.sort(np.random.default_rng(n).normal(size=n))
        lerp = x[k0] + (x[k1] - x[k0]) * g if g < 0.5 else x[k1] - (x[k1] - x[k0]) * (1 - g)
        assert lerp == np.percentile(x, q * 100)


def test_data_loader_matches_reference_golden(tmp_path, golden):
    """xanthos_amd.data_load.DataLoader against the reference's DataLoader (tests/golden/loader.npz, made by driving
    the reference with this package's ConfigReader): ha -> km2, the DRT map flattening (vectorize) incl. cells outside
    the maps, flow distance < 1000 m, negative velocities, nan_to_num policy per array (precipitation keeps NaN),
    tairprev = previous cell, region / country tables, calibration observations, future-mode channel storage."""
    import io
    import zipfile
    from xanthos_amd.data_load import DataLoader
    g = golden('loader')
    root = str(tmp_path)
    zipfile.ZipFile(io.BytesIO(g['tree_zip'].tobytes())).extractall(root)
    ini = os.path.join(root, str(g['ini_name']))
    text = open(ini).read().replace(str(g['old_root']), root)
    open(ini, 'w').write(text)
    s = ConfigReader(ini)
    assert str(s.HistFlag) == 'False' and s.ChStorageFile.endswith('ch_storage.npy')
    s.device_transforms = False                     # host-side nan_to_num, as the reference loader does it
    d = DataLoader(s)
    exact = ('area', 'coords', 'basin_ids', 'region_ids', 'country_ids', 'latitude', 'cL', 'beta', 'rslimit', 'ae',
             'be', 'Tminopen', 'Tminclose', 'VPDclose', 'VPDopen', 'RBLmin', 'RBLmax', 'rc', 'emiss', 'alpha', 'lai',
             'laimax', 'laimin', 'tair_load', 'TMIN_load', 'rhs_load', 'wind_load', 'rsds_load', 'rlds_load',
             'tairprev_load', 'lct_load', 'elev', 'precip', 'tmin', 'flow_dist', 'flow_dir', 'str_velocity',
             'instream_flow', 'chs_prev', 'cal_obs')
    for k in exact:
        got, ref = np.asarray(getattr(d, k)), g[k]
        assert got.shape == ref.shape or got.reshape(ref.shape).shape == ref.shape, k
        assert np.array_equal(got.reshape(ref.shape), ref, equal_nan=True), k
    for k in ('basin_names', 'region_names', 'country_names'):
        assert [str(x) for x in getattr(d, k)] == [str(x) for x in g[k]], k
    assert np.isnan(d.precip).any() and not np.isnan(d.tmin).any() and not np.isnan(d.tair_load).any()
    assert (d.flow_dist >= 1000).all() and (d.str_velocity >= 0).all() and d.chs_prev.max() > 0
    assert d.flow_dist[10] == 1000 and d.flow_dir[11] == -9999            # cells outside the maps: -9999 -> rep_val
    # device_transforms (the default): the big arrays keep their NaNs on the host and lose them on the device
    s2 = ConfigReader(ini)
    d2 = DataLoader(s2)
    assert np.isnan(d2.tair_load).any() and np.array_equal(np.nan_to_num(d2.tair_load), g['tair_load'])
    # ... and stay on disk: read-only memory maps of the .npy files (the pipeline sends the files' bytes to the GPU itself);
    # tairprev_load is only built when somebody asks for it, from the NaN-free temperatures like the reference's
    npy_backed = [k for k in ('tair_load', 'precip', 'rhs_load') if isinstance(getattr(d2, k), np.memmap)]
    assert len(npy_backed) == 3
    for k in npy_backed:
        mm = getattr(d2, k)
        assert not mm.flags.writeable and mm.dtype == np.float64 and os.path.isfile(mm.filename) and mm.offset > 0
    assert d2._tairprev is None
    assert np.array_equal(d2.tairprev_load, g['tairprev_load']) and d2._tairprev is not None
    s3 = ConfigReader(ini)
    s3.mmap_inputs = False                          # eager host arrays, writable, as the reference keeps them
    d3 = DataLoader(s3)
    assert not isinstance(d3.precip, np.memmap) and d3.precip.flags.writeable
    assert np.array_equal(d3.precip, d2.precip, equal_nan=True)


def test_histflag_is_normalised_and_future_mode_needs_channel_storage(tmp_path, golden, caplog, monkeypatch):
    """HistFlag in any of the reference's spellings (it compares the raw string three different ways: ini_reader.py:330,
    :582, data_load.py:431): historic iff it reads as true; future mode without ChStorageFile starts from empty channels like the
    reference, with a warning (an error under XH_STRICT_FUTURE=1); anything else is rejected."""
    import io
    import zipfile
    g = golden('loader')
    root = str(tmp_path)
    zipfile.ZipFile(io.BytesIO(g['tree_zip'].tobytes())).extractall(root)
    ini = os.path.join(root, str(g['ini_name']))
    text = open(ini).read().replace(str(g['old_root']), root)

    def reader(edit):
        path = os.path.join(root, 'edited.ini')
        open(path, 'w').write(edit(text))
        return ConfigReader(path)
    for spelling in ('false', 'F', 'no', '0', 'False'):
        s = reader(lambda t, sp=spelling: re.sub(r'(?m)^HistFlag\s*=.*$', 'HistFlag = ' + sp, t))
        assert s.HistFlag == 'False' and not s.historic and s.ChStorageFile.endswith('ch_storage.npy')
    for spelling in ('true', 'T', 'yes', '1'):
        s = reader(lambda t, sp=spelling: re.sub(r'(?m)^HistFlag\s*=.*$', 'HistFlag = ' + sp, t))
        assert s.HistFlag == 'True' and s.historic and s.ChStorageFile is None
    # the reference accepts future mode without ChStorageFile and starts from empty channels (data_load.py:427-438): a
    # warning here, an error only on request
    with caplog.at_level('WARNING'):
        s = reader(lambda t: re.sub(r'(?m)^ChStorageFile\s*=.*$', '', t))
    assert not s.historic and s.ChStorageFile is None and 'EMPTY channels' in caplog.text
    monkeypatch.setenv('XH_STRICT_FUTURE', '1')
    with pytest.raises(ValidationException, match='ChStorageFile'):
        reader(lambda t: re.sub(r'(?m)^ChStorageFile\s*=.*$', '', t))
    monkeypatch.delenv('XH_STRICT_FUTURE')
    with pytest.raises(ValidationException, match='HistFlag'):
        reader(lambda t: re.sub(r'(?m)^HistFlag\s*=.*$', 'HistFlag = maybe', t))


def test_route_planner_fuzz_under_sanitizers():
    """The routing planner is host code without any HIP in it (xanthos_amd/csrc/xh_flow_plan.cpp); tests/plan_fuzz builds
    that translation unit with -fsanitize=address,undefined and runs it over random forests, chains, stars, single cells,
    ~10^5-cell grids and inputs that are not trees (cycles, two downstream rows, missing diagonals), with and without a
    typed partition, under random planner options.  Every plan must pass flow_tables_check: every cell in exactly one
    slot, streams strictly down the pipeline, <= 16 imports / outlets per unit, even lags consistent with the 'two
    iterations earlier' rule, every row -- expanded through its chains -- equal to the CSR row in stored order
    (mrtm.py:50-51), plain units free of cells that need pairs."""
    import subprocess
    fuzz = os.path.join(ROOT, 'tests', 'plan_fuzz')
    subprocess.run(['make', '-C', fuzz], check=True, capture_output=True)
    out = subprocess.run([os.path.join(fuzz, 'plan_fuzz'), '250', '20240807'], capture_output=True, text=True, timeout=600)
    assert out.returncode == 0 and '250 cases, 0 failed' in out.stdout, out.stdout[-2000:] + out.stderr[-4000:]


def test_file_range_of_memory_map(tmp_path):
    """pipeline.file_range_of: where a float64 memory map starts in its file -- from the addresses, because a slice of a
    np.memmap keeps its parent's ``offset`` (the direct file -> HBM upload would otherwise send the wrong rows)."""
    from xanthos_amd.pipeline import file_range_of
    a = np.arange(5000 * 7, dtype=np.float64).reshape(5000, 7)
    path = str(tmp_path / 'a.npy')
    np.save(path, a)
    mm = np.load(path, mmap_mode='r')
    assert file_range_of(mm) == (path, mm.offset)
    sl = mm[1200:]
    assert sl.offset == mm.offset                          # numpy's pitfall
    fn, off = file_range_of(sl)
    raw = open(path, 'rb').read()
    assert np.array_equal(np.frombuffer(raw[off:off + sl.nbytes]).reshape(sl.shape), a[1200:])
    assert file_range_of(mm[:, :3]) is None                # not contiguous
    assert file_range_of(a) is None and file_range_of(np.asarray(mm)) is None
    np.save(str(tmp_path / 'f4.npy'), a.astype(np.float32))
    assert file_range_of(np.load(str(tmp_path / 'f4.npy'), mmap_mode='r')) is None


def test_bench_starts_its_own_ranks_without_a_launcher():
    """``python bench.py --gpus N`` with no RANK in the environment starts N rank processes itself (before anything touches a
    GPU: this test has none), gives each RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* as torch.distributed.run would, relays
    rank 0's line as its own stdout and exits with the worst exit code of its ranks."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ('RANK', 'LOCAL_RANK', 'WORLD_SIZE', 'MASTER_ADDR', 'MASTER_PORT')}
    for n, code in ((3, 0), (2, 7)):
        out = subprocess.run([sys.executable, os.path.join(root, 'bench.py'), '--gpus', str(n), '--launch-echo', str(code)],
                             env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=120)
        assert out.returncode == code, (out.returncode, out.stderr[-500:])
        got = json.loads(out.stdout.strip().splitlines()[-1])
        assert got['RANK'] == '0' and got['LOCAL_RANK'] == '0' and got['WORLD_SIZE'] == str(n)
        assert got['MASTER_ADDR'] == '127.0.0.1' and int(got['MASTER_PORT']) > 0
