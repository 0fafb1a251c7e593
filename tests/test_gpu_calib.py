"""GPU tests (-m gpu) of the device-side differential evolution and of BASELINE config 5 at its full size.

The device DE (csrc/xh_calib_de.hip) is held to oracle/de.py -- the numpy restatement of SciPy's best1bin generation
that tests/test_oracle_de.py pins to SciPy's own solver -- and the objective to oracle/calib.py (pinned by
tests/golden/kge.npz).  Everything goes through the C-ABI (xanthos_amd/_hip.py).
"""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

BOUNDS = [(1e-4, 1 - 1e-4), (1e-4, 8 - 1e-4), (1e-4, 1 - 1e-4), (1e-4, 1 - 1e-4), (1e-4, 1 - 1e-4)]


@pytest.fixture(scope='module')
def hip():
    from xanthos_amd import _hip
    return _hip


@pytest.fixture(scope='module')
def small(hip):
    """Three basins of a 900-cell world, 36 months, observations made by the oracle at known parameters."""
    from oracle import calib as o_calib
    from xanthos_amd import synth
    from xanthos_amd.calibrate.calibrate_abcd import BasinSet, Calibrate
    w = synth.make_world(nrow=36, ncol=72, ncell=900, n_basins=7, seed=33)
    nm, spin = 36, 25
    f = synth.make_forcing(w, nm)
    pet = np.random.default_rng(1).uniform(20, 150, (w.ncell, nm))
    truth = np.array([0.96, 0.8, 0.5, 0.4, 0.3])
    rows = []
    for b in (1, 2, 3):
        sel = w.basin_ids == b
        series = o_calib.basin_runoff(truth, 0, pet[sel], f['precip'][sel], f['abcd_tmin'][sel], nm, spin,
                                      'km3_per_mth', w.area[sel])
        rows.append(np.stack([np.full(nm, b), series], axis=1))
    obs = np.concatenate(rows)
    cals = [Calibrate(basin_num=b, basin_ids=w.basin_ids, basin_areas=w.area, precip=f['precip'], pet=pet, obs=obs,
                      tmin=f['abcd_tmin'], n_months=nm, runoff_spinup=spin, set_calibrate=0, obs_unit='km3_per_mth',
                      out_dir=None) for b in (1, 2, 3)]
    bset = BasinSet(cals, nm, spin, 'km3_per_mth')
    yield bset, w, f, pet, obs
    bset.close()


def test_de_generation_matches_oracle(small):
    """Initial population, trial vectors and scaled parameters bit for bit; selection and convergence exactly."""
    from oracle import de as o_de
    bset = small[0]
    seed, n, d = 4242, 40, 5
    lo, hi = np.array([b[0] for b in BOUNDS]), np.array([b[1] for b in BOUNDS])
    de = bset.solver(n, seed=seed)
    try:
        de.init()
        pop, en = de.state(0)
        keys = [c.basin_num for c in bset.cals]
        for b, key in enumerate(keys):
            assert np.array_equal(pop[b], o_de.init_population(seed, key, n, d))
        # energies of the initial population = the objective of the scaled vectors (same kernel, same numbers)
        assert np.array_equal(en, o_de.clean(bset.evaluate(o_de.scale_parameters(pop, lo, hi))))
        for gen in range(6):
            left = de.step(1, tol=0.01)
            trial, e_trial = de.state(1)
            scaled, _ = de.state(2)
            new_pop, new_en = de.state(0)
            for b, key in enumerate(keys):
                want = o_de.generation_trial(seed, key, gen, pop[b], en[b])
                assert np.array_equal(trial[b], want), (gen, b)
                assert np.array_equal(scaled[b], o_de.scale_parameters(want, lo, hi))
                p2, e2 = o_de.select(pop[b], en[b], want, e_trial[b])
                assert np.array_equal(new_pop[b], p2) and np.array_equal(new_en[b], e2)
            assert np.array_equal(e_trial, bset.evaluate(scaled))
            assert left == 3                                   # far from converged after a few generations
            pop, en = new_pop, new_en
        x, fun, nfev, nit, act = de.result()
        assert (nfev == n * 7).all() and (nit == 6).all() and act.all()
        k = np.argmin(en, axis=1)
        assert np.array_equal(fun, en[np.arange(3), k])
        assert np.array_equal(x, o_de.scale_parameters(pop[np.arange(3), k], lo, hi))
    finally:
        de.close()


def test_de_search_equals_oracle_search_and_recovers_truth(small):
    """A whole search on the device = the oracle's search driven by the same objective: same generations, same
    result.  Basins converge at different generations and drop out of the launches one by one."""
    from oracle import de as o_de
    from xanthos_amd.calibrate.calibrate_abcd import differential_evolution_device
    bset = small[0]
    seed, n = 77, 50
    x, fun, nfev, nit = differential_evolution_device(bset, seed=seed, nmembers=n, check_every=1)
    assert (fun < 0.02).all() and (nit > 5).all() and (nfev == n * (nit + 1)).all()
    for b, c in enumerate(bset.cals):
        one = lambda X, b=b: bset.evaluate(np.repeat(X[None], len(bset.cals), axis=0))[b]
        ox, ofun, onfev, onit = o_de.differential_evolution(one, BOUNDS, seed, c.basin_num, n)
        assert (onit, onfev) == (int(nit[b]), int(nfev[b])), (b, onit, nit[b])
        assert np.array_equal(ox, x[b]) and ofun == fun[b]
    # check_every > 1: a basin that converges inside a batch of generations is frozen at once
    x4, fun4, nfev4, nit4 = differential_evolution_device(bset, seed=seed, nmembers=n, check_every=5)
    assert np.array_equal(x4, x) and np.array_equal(nit4, nit) and np.array_equal(nfev4, nfev)
    # a basin's search does not depend on its companions (counter-based streams keyed on the basin number)
    from xanthos_amd.calibrate.calibrate_abcd import BasinSet
    alone = BasinSet([bset.cals[1]], bset.n_months, bset.spinup, 'km3_per_mth')
    try:
        xa, fa, _, na = differential_evolution_device(alone, seed=seed, nmembers=n)
    finally:
        alone.close()
    assert np.array_equal(xa[0], x[1]) and fa[0] == fun[1] and na[0] == nit[1]


def test_de_frozen_basin_and_failed_members(small):
    """Converged basins are skipped by every kernel; NaN / inf energies never win and block convergence."""
    bset = small[0]
    n = 20
    de = bset.solver(n, seed=5)
    try:
        de.init()
        pop, en = de.state(0)
        en2 = en.copy()
        en2[0, :] = 1.0                       # basin 0: zero spread -> converges at the first test
        en2[1, 3] = np.inf                    # basin 1: an infinite energy blocks convergence
        en2[1, :3] = 1e-9                     # ... whatever the others are
        de.set_state(pop, en2, generation=0)
        left = de.step(1)
        x, fun, nfev, nit, act = de.result()
        assert act.tolist()[0] in (0, 1)
        p1, e1 = de.state(0)
        if act[0] == 0:
            left2 = de.step(3)
            p2, e2 = de.state(0)
            assert np.array_equal(p2[0], p1[0]) and np.array_equal(e2[0], e1[0])      # frozen
            assert de.result()[3][0] == 1 and left2 <= left
        assert np.isfinite(e1[2]).all()
    finally:
        de.close()


def test_calibration_tmin_nan_to_num(small, hip):
    """TempMinFile with missing values: the loader's nan_to_num (data_load.py:194-195) happens on the device inside the
    calibration objective (round-1 advisor finding): NaN -> 0 (all snow), +/-inf -> +/-largest double."""
    from oracle import calib as o_calib
    from xanthos_amd.calibrate.calibrate_abcd import objective_kge
    bset, w, f, pet, obs = small
    sel = w.basin_ids == 2
    tmin = f['abcd_tmin'][sel].copy()
    rng = np.random.default_rng(3)
    tmin[rng.random(tmin.shape) < 0.05] = np.nan
    tmin[0, 3], tmin[1, 4] = np.inf, -np.inf
    robs = obs[obs[:, 0] == 2][:, 1]
    for pars in ([0.9, 1.2, 0.4, 0.3, 0.5], [0.5, 4.0, 0.8, 0.1, 0.9]):
        got = objective_kge(np.array(pars), pet[sel], f['precip'][sel], tmin, 36, 25, 'km3_per_mth', w.area[sel], robs)
        ref = o_calib.objective_kge(np.array(pars), 0, pet[sel], f['precip'][sel], np.nan_to_num(tmin), 36, 25,
                                    'km3_per_mth', w.area[sel], robs)
        raw = o_calib.objective_kge(np.array(pars), 0, pet[sel], f['precip'][sel], tmin, 36, 25, 'km3_per_mth',
                                    w.area[sel], robs)
        assert abs(got - ref) <= 1e-9 * abs(ref) and abs(raw - ref) > 1e-6


def test_config5_full_size_generation(hip):
    """BASELINE config 5 at its workload: 512 members x 235 basins x (480 + 120) months of the 67,420-cell world in
    one device-side DE generation; a sample of (basin, member) objectives against oracle/calib.py, the trial vectors
    of a sample of basins against oracle/de.py, and the host share of a generation's wall time."""
    import time
    from oracle import calib as o_calib, de as o_de
    from xanthos_amd import synth
    from xanthos_amd.calibrate.config5 import Config5
    cfg = Config5(hip.get_context(0), nmembers=512, nmonths=480, spinup=120, seed=9)
    try:
        de, w = cfg.de, cfg.world
        de.init()
        pop, en = de.state(0)
        assert pop.shape == (w.n_basins, 512, 5) and np.isfinite(en).all()
        lo, hi = np.array([b[0] for b in BOUNDS]), np.array([b[1] for b in BOUNDS])
        rng = np.random.default_rng(0)
        sample = [(int(b), int(m)) for b, m in zip(rng.integers(0, w.n_basins, 10), rng.integers(0, 512, 10))]
        sample += [(int(np.argmax(cfg.counts)), 0), (int(np.argmin(cfg.counts)), 511)]
        worst = 0.0
        for b, m in sample:
            host = cfg.host_basin(b)
            ref = o_calib.objective_kge(o_de.scale_parameters(pop[b, m], lo, hi), 0, host['pet'], host['precip'],
                                        host['tmin'], 480, 120, 'km3_per_mth', host['area'], cfg.obs[b])
            worst = max(worst, abs(en[b, m] - ref) / abs(ref))
        assert worst <= 1e-9, worst
        ctx = hip.get_context(0)
        ctx.timing_reset()
        t0 = time.perf_counter()
        left = de.step(3)
        wall = (time.perf_counter() - t0) / 3
        kernels = sum(ctx.timing(k)[0] for k in ('calib_abcd', 'calib_kge', 'calib_de')) / 3e3
        trial, e_trial = de.state(1)
        for b in (0, 117, 234):
            # generation 2's trial vectors from the population before it: replay the three generations on the host
            p, e = pop[b], en[b]
            for gen in range(3):
                t = o_de.generation_trial(9, cfg.keys[b], gen, p, e)
                if gen == 2:
                    assert np.array_equal(trial[b], t)
                else:
                    one = cfg.evaluate_one(b, o_de.scale_parameters(t, lo, hi))
                    p, e = o_de.select(p, e, t, one)
        assert left == w.n_basins
        print('config 5 generation: wall %.4f s, kernels %.4f s' % (wall, kernels))
        assert wall <= 1.5 * kernels, (wall, kernels)        # the host adds nothing to a generation
    finally:
        cfg.close()
