"""Differential evolution around the calibration objective (oracle; test infrastructure only).

The reference calls ``scipy.optimize.differential_evolution(objective_kge, bounds, popsize=15, polish=False)``
(xanthos/calibrate/calibrate_abcd.py:103-112) with SciPy's defaults: strategy ``best1bin``, Latin-hypercube start,
mutation dithered in (0.5, 1) once per generation, recombination 0.7, ``tol = 0.01``, ``maxiter = 1000``.  SciPy is a
third-party dependency (setup pin ``scipy>=1.6``; 1.15.3 in this image) and its generator is unseeded there, so the
reference's search TRAJECTORY is not reproducible: "parity unpinned" for the trajectory.  What can be pinned is

* the objective (``oracle.calib``, pinned by tests/golden/kge.npz), and
* the generation step as an algorithm: this file restates SciPy's ``best1bin`` generation (``_mutate`` / ``_best1`` /
  ``_select_samples`` / ``_ensure_constraint`` / ``_scale_parameters`` / ``converged`` of
  scipy/optimize/_differentialevolution.py) with ``updating='deferred'`` semantics, drawing its random numbers from
  the same counter-based SplitMix64 streams as the device kernels (xanthos_amd/csrc/xh_calib_de.hip), so that the
  device's trial vectors can be compared with it bit for bit and its selection / convergence decisions exactly.

``tests/test_oracle_de.py`` additionally checks the step against SciPy's own solver: fed SciPy's random draws, the
restated step reproduces ``DifferentialEvolutionSolver``'s trial vectors.
"""
import numpy as np

_U64 = np.uint64
SLOT_R0, SLOT_R1, SLOT_FILL, SLOT_SCALE, SLOT_CROSS, SLOT_REDRAW = 0, 1, 2, 3, 8, 40
MEMBER_GEN = 0x7fffffff


def _splitmix64(x):
    with np.errstate(over='ignore'):
        z = np.asarray(x, dtype=_U64) + _U64(0x9E3779B97F4A7C15)
        z = (z ^ (z >> _U64(30))) * _U64(0xBF58476D1CE4E5B9)
        z = (z ^ (z >> _U64(27))) * _U64(0x94D049BB133111EB)
        return z ^ (z >> _U64(31))


def de_uniform(seed, key, gen, member, slot):
    """U[0,1) for (basin key, generation, member, slot); generation -1 = initial population. Broadcasts."""
    with np.errstate(over='ignore'):
        h = _splitmix64(_U64(seed) ^ (np.asarray(key, dtype=_U64) * _U64(0xD1342543DE82EF95)))
        h = _splitmix64(h ^ np.asarray(np.asarray(gen, dtype=np.int64) + 1, dtype=np.uint32).astype(_U64))
        ms = (np.asarray(member, dtype=np.int64).astype(np.uint32).astype(_U64) << _U64(32)) | \
            np.asarray(slot, dtype=np.int64).astype(np.uint32).astype(_U64)
        h = _splitmix64(h ^ ms)
    return (h >> _U64(11)).astype(np.float64) * (1.0 / 9007199254740992.0)


def scale_parameters(t, lo, hi):
    """SciPy ``_scale_parameters``: 0.5 (lo + hi) + (t - 0.5) |lo - hi|."""
    lo, hi = np.asarray(lo, dtype=float), np.asarray(hi, dtype=float)
    return 0.5 * (lo + hi) + (t - 0.5) * np.abs(lo - hi)


def init_population(seed, key, n, d):
    """Latin hypercube start (SciPy ``init_population_lhs``): per gene a random permutation of the n strata + jitter."""
    i = np.arange(n)
    pop = np.empty((n, d))
    for j in range(d):
        k = de_uniform(seed, key, -1, i, SLOT_CROSS + j)
        rank = np.empty(n, dtype=np.int64)
        rank[np.lexsort((i, k))] = i                       # rank of k[i], ties by index
        pop[:, j] = (rank + de_uniform(seed, key, -1, i, SLOT_REDRAW + j)) / n
    return pop


def select_samples(u0, u1, i, n):
    """Two distinct members, both different from candidate ``i``, uniformly (SciPy ``_select_samples``)."""
    r0 = np.minimum((u0 * (n - 1)).astype(np.int64), n - 2)
    r0 = r0 + (r0 >= i)
    r1 = np.minimum((u1 * (n - 2)).astype(np.int64), n - 3)
    s0, s1 = np.minimum(i, r0), np.maximum(i, r0)
    r1 = r1 + (r1 >= s0)
    r1 = r1 + (r1 >= s1)
    return r0, r1


def best1bin_trial(pop, energies, scale, r0, r1, cross, fill, redraw):
    """One ``best1bin`` generation of trial vectors from explicit random draws.

    pop [n, d] in the unit cube, energies [n]; scale scalar; r0, r1 [n] member indices; cross [n, d] booleans
    (uniform < recombination); fill [n] forced gene; redraw [n, d] replacement values for out-of-bounds genes."""
    n, d = pop.shape
    best = pop[int(np.argmin(energies))]
    mutant = best[None, :] + scale * (pop[r0] - pop[r1])               # _best1
    cross = cross.copy()
    cross[np.arange(n), fill] = True                                    # binomial crossover with one forced gene
    trial = np.where(cross, mutant, pop)
    out = (trial < 0) | (trial > 1) | np.isnan(trial)                   # _ensure_constraint
    return np.where(out, redraw, trial)


def generation_trial(seed, key, gen, pop, energies, mutation=(0.5, 1.0), recombination=0.7):
    """Trial vectors of generation ``gen`` for one basin, with the device's random streams."""
    n, d = pop.shape
    i = np.arange(n)
    scale = mutation[0] + de_uniform(seed, key, gen, MEMBER_GEN, SLOT_SCALE) * (mutation[1] - mutation[0])
    r0, r1 = select_samples(de_uniform(seed, key, gen, i, SLOT_R0), de_uniform(seed, key, gen, i, SLOT_R1), i, n)
    fill = np.minimum((de_uniform(seed, key, gen, i, SLOT_FILL) * d).astype(np.int64), d - 1)
    jj = np.arange(d)[None, :]
    cross = de_uniform(seed, key, gen, i[:, None], SLOT_CROSS + jj) < recombination
    redraw = de_uniform(seed, key, gen, i[:, None], SLOT_REDRAW + jj)
    return best1bin_trial(pop, energies, float(scale), r0, r1, cross, fill, redraw)


def clean(e):
    e = np.asarray(e, dtype=float)
    return np.where(np.isfinite(e), e, np.inf)


def select(pop, energies, trial, e_trial):
    """Deferred selection: a trial replaces its member when its energy is <= the member's."""
    e_trial = clean(e_trial)
    better = e_trial <= energies
    return np.where(better[:, None], trial, pop), np.where(better, e_trial, energies)


def converged(energies, tol=0.01, atol=0.0):
    """SciPy ``converged()``: never with an infinite energy; std <= atol + tol |mean|."""
    if np.any(np.isinf(energies)):
        return False
    return bool(np.std(energies) <= atol + tol * np.abs(np.mean(energies)))


def differential_evolution(func, bounds, seed, key, nmembers, maxiter=1000, tol=0.01, atol=0.0, mutation=(0.5, 1.0),
                           recombination=0.7):
    """Whole search for one basin. ``func(X[n, d]) -> energies[n]``. Returns (x, fun, nfev, nit)."""
    lo = np.array([b[0] for b in bounds], dtype=float)
    hi = np.array([b[1] for b in bounds], dtype=float)
    d = len(bounds)
    pop = init_population(seed, key, nmembers, d)
    energies = clean(func(scale_parameters(pop, lo, hi)))
    nfev, nit = nmembers, 0
    for gen in range(maxiter):
        trial = generation_trial(seed, key, gen, pop, energies, mutation, recombination)
        pop, energies = select(pop, energies, trial, func(scale_parameters(trial, lo, hi)))
        nfev += nmembers
        nit += 1
        if converged(energies, tol, atol):
            break
    k = int(np.argmin(energies))
    return scale_parameters(pop[k], lo, hi), float(energies[k]), nfev, nit
