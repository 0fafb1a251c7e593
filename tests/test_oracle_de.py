"""CPU tests that pin oracle/de.py (the numpy restatement of SciPy's best1bin generation) to SciPy's own solver.

The reference calls scipy.optimize.differential_evolution (calibrate_abcd.py:103-112).  SciPy is importable here, so the
restated generation step is checked against ``DifferentialEvolutionSolver`` itself: fed the random draws SciPy makes
(a twin generator with the same seed, consumed in the order SciPy 1.15 consumes it), the oracle's trial vectors must
equal SciPy's exactly.  The device kernels are then held to the oracle in tests/test_gpu_calib.py.
"""
import numpy as np
import pytest
import scipy
from scipy.optimize._differentialevolution import DifferentialEvolutionSolver

from oracle import de as o_de

BOUNDS = [(1e-4, 1 - 1e-4), (1e-4, 8 - 1e-4), (1e-4, 1 - 1e-4), (1e-4, 1 - 1e-4), (1e-4, 1 - 1e-4)]


def _quadratic(x):
    return float(np.sum((np.asarray(x) - np.array([0.3, 5.0, 0.7, 0.2, 0.6])) ** 2))


def _solver(seed, popsize=6):
    return DifferentialEvolutionSolver(_quadratic, BOUNDS, popsize=popsize, polish=False, updating='deferred',
                                       rng=np.random.default_rng(seed))


@pytest.mark.skipif(tuple(int(v) for v in scipy.__version__.split('.')[:2]) < (1, 12),
                    reason='the twin-generator replay follows the draw order of SciPy >= 1.12 (_mutate_many)')
@pytest.mark.parametrize('seed', [1, 2, 3])
def test_trial_vectors_equal_scipys(seed):
    s = _solver(seed)
    n, d = s.population.shape
    rng = np.random.default_rng(99 + seed)
    s.population_energies = rng.uniform(0.5, 2.0, n)
    s._promote_lowest_energy()                       # SciPy keeps the best member at index 0
    s.scale = 0.77
    # make some mutants leave the unit cube so that the re-draw path is exercised
    s.population[3] = [0.99, 0.01, 0.98, 0.02, 0.5]
    s.population[4] = [0.01, 0.99, 0.02, 0.98, 0.5]
    pop, en = s.population.copy(), s.population_energies.copy()
    assert int(np.argmin(en)) == 0

    # twin generator: replay SciPy's draws (_select_samples shuffles a persistent index array once per candidate,
    # then one integers() call for the forced genes and one uniform() call for the crossover mask)
    twin = np.random.default_rng()
    twin.bit_generator.state = s.random_number_generator.bit_generator.state
    idx = s._random_population_index.copy()
    r0, r1 = np.empty(n, dtype=np.int64), np.empty(n, dtype=np.int64)
    for c in range(n):
        twin.shuffle(idx)
        smp = idx[:6][idx[:6] != c][:5]
        r0[c], r1[c] = smp[0], smp[1]
    fill = twin.integers(0, d, size=n)
    cross = twin.uniform(size=(n, d)) < s.cross_over_probability

    got_scipy = s._mutate_many(np.arange(n))
    out = (got_scipy < 0) | (got_scipy > 1)
    assert out.any(), 'test should exercise out-of-bounds genes'
    redraw = np.full((n, d), 0.123456)
    mine = o_de.best1bin_trial(pop, en, s.scale, r0, r1, cross, fill, redraw)
    assert np.array_equal(mine[~out], got_scipy[~out])               # bit for bit
    assert np.all(mine[out] == 0.123456)                             # _ensure_constraint: uniform re-draw
    for t in got_scipy:
        s._ensure_constraint(t)
    assert ((got_scipy >= 0) & (got_scipy <= 1)).all()
    # the samples SciPy draws never include the candidate and are distinct
    assert (r0 != np.arange(n)).all() and (r1 != np.arange(n)).all() and (r0 != r1).all()


def test_scaling_and_convergence_equal_scipys():
    s = _solver(5)
    lo, hi = np.array([b[0] for b in BOUNDS]), np.array([b[1] for b in BOUNDS])
    t = np.random.default_rng(0).random((7, 5))
    assert np.array_equal(o_de.scale_parameters(t, lo, hi), s._scale_parameters(t))
    for e in (np.full(30, 2.0), np.linspace(1.0, 1.02, 30), np.linspace(1.0, 1.2, 30),
              np.r_[np.ones(29), np.inf]):
        s.population_energies = e.copy()
        assert o_de.converged(e, s.tol, s.atol) == bool(s.converged())


def test_select_samples_uniform_and_distinct():
    n = 7
    rng = np.random.default_rng(4)
    counts = np.zeros((n, n, n), dtype=np.int64)
    for _ in range(300):
        i = np.arange(n)
        r0, r1 = o_de.select_samples(rng.random(n), rng.random(n), i, n)
        assert (r0 != i).all() and (r1 != i).all() and (r0 != r1).all()
        assert ((r0 >= 0) & (r0 < n) & (r1 >= 0) & (r1 < n)).all()
        np.add.at(counts, (i, r0, r1), 1)
    # every ordered pair (r0, r1) outside the candidate is reachable and roughly equally likely: 30 pairs, 300 draws
    for i in range(n):
        c = counts[i][np.ix_([k for k in range(n) if k != i], [k for k in range(n) if k != i])]
        off = c[~np.eye(n - 1, dtype=bool)]
        assert off.min() >= 1 and off.max() <= 30
    # edge values of the uniforms never index out of range
    u = np.array([0.0, np.nextafter(1.0, 0.0)])
    for i in range(4):
        for a in u:
            for b in u:
                r0, r1 = o_de.select_samples(np.array([a]), np.array([b]), np.array([i]), 4)
                assert {int(r0[0]), int(r1[0]), i} <= set(range(4)) and len({int(r0[0]), int(r1[0]), i}) == 3


def test_latin_hypercube_start():
    n, d = 25, 5
    pop = o_de.init_population(123, 17, n, d)
    assert pop.shape == (n, d) and ((pop >= 0) & (pop < 1)).all()
    for j in range(d):                                  # one member per stratum in every gene
        assert sorted(np.floor(pop[:, j] * n).astype(int)) == list(range(n))
    assert not np.array_equal(np.argsort(pop[:, 0]), np.argsort(pop[:, 1]))      # genes are permuted independently
    assert not np.array_equal(pop, o_de.init_population(123, 18, n, d))          # another basin key, another start
    assert np.array_equal(pop, o_de.init_population(123, 17, n, d))


def test_whole_search_finds_minimum_like_scipy():
    target = np.array([0.3, 5.0, 0.7, 0.2, 0.6])
    f = lambda X: np.sum((X - target) ** 2, axis=1)
    x, fun, nfev, nit = o_de.differential_evolution(f, BOUNDS, seed=7, key=3, nmembers=75)
    assert np.allclose(x, target, atol=5e-2) and fun < 1e-2 and nfev == 75 * (nit + 1)
    res = scipy.optimize.differential_evolution(_quadratic, BOUNDS, popsize=15, polish=False, updating='deferred',
                                                rng=1)
    assert np.allclose(res.x, target, atol=5e-2)
    assert 0.3 < nit / res.nit < 3.0                    # same algorithm, same order of effort
