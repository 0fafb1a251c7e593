"""ABCD calibration on MI355X -- mirror of xanthos/calibrate/calibrate_abcd.py.

The reference wraps ``scipy.optimize.differential_evolution`` around ``objective_kge`` (:103-112, :176-213), which
runs ABCD on one basin for ONE parameter vector per call.  Here the objective is evaluated for a whole population
per call by ``xh_calib_objective`` (csrc/xh_calib.hip: members x cells on the GPU, forcing transposed once per basin
and kept in HBM), and the differential-evolution driver is a generation-synchronous ``best1bin`` with the same
defaults as SciPy's (popsize 15 x n_parameters members, Latin-hypercube start, dithered mutation in (0.5, 1),
recombination 0.7, tol 0.01, maxiter 1000).  SciPy's driver is unseeded and updates the population in place, so the
reference's search trajectory is not reproducible; only the objective is a parity target (tests/golden/kge.npz).

``set_calibrate = 1`` (calibrate against routed stream flow) is not offered: in the reference that branch hands the
whole ``[ncell, nmonths]`` Avg_ChFlow array to ``np.corrcoef`` against a 1-D observation series (:165-173, :196-213),
which cannot produce a basin score.
"""
import logging
import os
import time

import numpy as np

from .. import _hip

LB = 1e-4
UB = 1 - LB


class BasinObjective:
    """ED = 1 - KGE of the basin runoff for batches of parameter vectors (basin_runoff + objective_kge, :134-213)."""

    def __init__(self, pet, precip, tmin, n_months, runoff_spinup, obs_unit, bsn_areas, bsn_robs, device=0):
        if obs_unit not in ('km3_per_mth', 'mm_per_mth'):
            raise ValueError('obs_unit must be km3_per_mth or mm_per_mth for set_calibrate = 0')
        if runoff_spinup < 25:
            raise IndexError('Spin-up steps must produce at least 25 months of spin-up; got {}'.format(runoff_spinup))
        self.ctx = _hip.get_context(device)
        self.ncell = int(np.asarray(pet).shape[0])
        self.n_months, self.spinup = int(n_months), int(runoff_spinup)
        self.nosnow = tmin is None
        self.npar = 4 if self.nosnow else 5
        tr = lambda a: self.ctx.upload(np.ascontiguousarray(np.asarray(a, dtype=np.float64)[:, :n_months].T))
        self.d_pet, self.d_precip = tr(pet), tr(precip)
        self.d_tmin = None if self.nosnow else tr(tmin)
        self.d_area = self.ctx.upload(bsn_areas) if obs_unit == 'km3_per_mth' else None
        self.obs = np.ascontiguousarray(np.asarray(bsn_robs, dtype=np.float64)[:n_months])
        self.nfev = 0

    def __call__(self, pars, want_series=False):
        pars = np.atleast_2d(np.asarray(pars, dtype=np.float64))[:, :self.npar]
        self.nfev += pars.shape[0]
        return self.ctx.calib_objective(self.ncell, self.n_months, self.spinup, pars, self.d_pet, self.d_precip,
                                        self.d_tmin, self.d_area, self.obs, want_series=want_series)

    def close(self):
        for b in (self.d_pet, self.d_precip, self.d_tmin, self.d_area):
            if b is not None:
                b.free()


def differential_evolution_batched(func, bounds, popsize=15, maxiter=1000, tol=0.01, atol=0.0, mutation=(0.5, 1.0),
                                   recombination=0.7, seed=None):
    """Generation-synchronous DE/best/1/bin. ``func(P[n, d]) -> energies[n]``. Returns (x, fun, nfev, nit)."""
    rng = np.random.default_rng(seed)
    lo = np.array([b[0] for b in bounds], dtype=float)
    hi = np.array([b[1] for b in bounds], dtype=float)
    d = len(bounds)
    n = max(5, popsize * d)
    # Latin hypercube start, as SciPy's init='latinhypercube'
    seg = (np.arange(n)[:, None] + rng.random((n, d))) / n
    pop = np.empty((n, d))
    for j in range(d):
        pop[:, j] = seg[rng.permutation(n), j]
    energies = np.asarray(func(lo + pop * (hi - lo)), dtype=float)
    energies = np.where(np.isfinite(energies), energies, np.inf)
    nfev, nit = n, 0
    for nit in range(1, maxiter + 1):
        finite = energies[np.isfinite(energies)]
        if finite.size == n and np.std(finite) <= atol + tol * np.abs(np.mean(finite)):
            break
        best = pop[np.argmin(energies)]
        scale = rng.uniform(mutation[0], mutation[1])                  # dither once per generation
        r = np.array([rng.choice(n, 2, replace=False) for _ in range(n)])
        mutant = best + scale * (pop[r[:, 0]] - pop[r[:, 1]])
        cross = rng.random((n, d)) < recombination
        cross[np.arange(n), rng.integers(0, d, n)] = True
        trial = np.where(cross, mutant, pop)
        out = (trial < 0) | (trial > 1)
        trial[out] = rng.random(int(out.sum()))                        # SciPy re-draws out-of-bounds entries
        e_trial = np.asarray(func(lo + trial * (hi - lo)), dtype=float)
        e_trial = np.where(np.isfinite(e_trial), e_trial, np.inf)
        nfev += n
        better = e_trial <= energies
        pop[better], energies[better] = trial[better], e_trial[better]
    k = int(np.argmin(energies))
    return lo + pop[k] * (hi - lo), float(energies[k]), nfev, nit


class Calibrate:
    """Calibrate the ABCD runoff module for one basin; constructor as calibrate_abcd.Calibrate (:20-88)."""

    def __init__(self, basin_num, basin_ids, basin_areas, precip, pet, obs, tmin, n_months, runoff_spinup,
                 set_calibrate, obs_unit, out_dir, router_func=None, device=0, seed=None):
        if set_calibrate != 0:
            raise NotImplementedError('set_calibrate = 1 (stream flow) is not supported; see the module docstring')
        self.basin_num, self.n_months, self.runoff_spinup = basin_num, n_months, runoff_spinup
        self.set_calibrate, self.obs_unit, self.out_dir, self.seed = set_calibrate, obs_unit, out_dir, seed
        self.nosnow = tmin is None
        self.bounds = [(LB, UB), (LB, 8 - LB), (LB, UB), (LB, UB), (LB, UB)]          # :62-64
        if self.nosnow:
            self.bounds.pop()
        self.all_pars = np.zeros((1, len(self.bounds)))
        self.kge_vals = np.zeros(1)
        self.basin_idx = np.where(np.asarray(basin_ids) == basin_num)
        self.bsn_areas = np.asarray(basin_areas)[self.basin_idx]
        self.bsn_PET = np.asarray(pet)[self.basin_idx]
        self.bsn_P = np.asarray(precip)[self.basin_idx]
        self.bsn_TMIN = None if self.nosnow else np.asarray(tmin)[self.basin_idx]
        obs = np.asarray(obs)
        self.bsn_Robs = obs[np.where(obs[:, 0] == basin_num)][:n_months, 1]           # :88
        self.device = device
        self.nfev = 0

    def objective(self):
        return BasinObjective(self.bsn_PET, self.bsn_P, self.bsn_TMIN, self.n_months, self.runoff_spinup,
                              self.obs_unit, self.bsn_areas, self.bsn_Robs, device=self.device)

    def calibrate_basin(self, popsize=15, polish=False):
        """Optimise (a, b, c, d[, m]) for maximum KGE and save the results (:90-131)."""
        st = time.time()
        obj = self.objective()
        try:
            x, ed, nfev, nit = differential_evolution_batched(obj, self.bounds, popsize=popsize, seed=self.seed)
        finally:
            obj.close()
        self.all_pars[0, :] = x
        self.kge_vals[0] = 1 - ed
        self.nfev = nfev
        par_names = 'abcd' + 'm' * (not self.nosnow)
        logging.debug('\t\tFinished calibration for basin {0} which contains {1} grid cells.'.format(
            self.basin_num, self.basin_idx[0].shape[0]))
        logging.debug('\t\tParameter values ({}):  {}'.format(','.join(list(par_names)), x))
        logging.debug('\t\tKGE:  {}'.format(1 - ed))
        logging.debug('\t\tNumber of function evaluations:  {} in {} generations'.format(nfev, nit))
        logging.debug('\t\tCalibration time (seconds):  {}'.format(time.time() - st))
        if self.out_dir is not None:
            os.makedirs(self.out_dir, exist_ok=True)
            np.save('{}/kge_result_basin_{}.npy'.format(self.out_dir, self.basin_num), self.kge_vals)
            np.save('{}/{}_parameters_basin_{}.npy'.format(self.out_dir, par_names, self.basin_num), self.all_pars)


def objective_kge(pars, pet, precip, tmin, n_months, runoff_spinup, obs_unit, bsn_areas, bsn_robs, device=0):
    """Single evaluation of the reference's objective_kge (:176-213) for set_calibrate = 0."""
    obj = BasinObjective(pet, precip, tmin, n_months, runoff_spinup, obs_unit, bsn_areas, bsn_robs, device=device)
    try:
        return float(obj(np.asarray(pars)[None, :])[0])
    finally:
        obj.close()


def expand_str_range(str_ranges):
    """['0-2', '6'] -> [0, 1, 2, 6] (:235-253)."""
    out = []
    for r in str_ranges:
        if '-' in r:
            a, b = r.split('-')
            out.extend(range(int(a), int(b) + 1))
        else:
            out.append(int(r))
    return out


def process_basin(basin_num, settings, data, pet, router_function=None):
    cal = Calibrate(basin_num=basin_num, set_calibrate=settings.set_calibrate, obs_unit=settings.obs_unit,
                    basin_ids=data.basin_ids, basin_areas=data.area, precip=data.precip, pet=pet, obs=data.cal_obs,
                    tmin=data.tmin, n_months=settings.nmonths, runoff_spinup=settings.runoff_spinup,
                    router_func=router_function, out_dir=settings.calib_out_dir,
                    device=getattr(settings, 'device', 0))
    cal.calibrate_basin()
    return cal


def differential_evolution_multi(func, bounds, nbasins, popsize=15, maxiter=1000, tol=0.01, atol=0.0,
                                 mutation=(0.5, 1.0), recombination=0.7, seed=None):
    """Lock-step DE/best/1/bin for ``nbasins`` independent problems with a shared generation clock.

    ``func(active, P)``: ``active`` = indices of the basins still searching, ``P`` [len(active), n, d] their trial
    populations; returns energies [len(active), n].  One call = one generation of every active basin, which is what
    the multi-basin GPU objective evaluates in a single launch.  Returns (x [nb, d], fun [nb], nfev [nb], nit [nb]).
    """
    rng = np.random.default_rng(seed)
    lo = np.array([b[0] for b in bounds], dtype=float)
    hi = np.array([b[1] for b in bounds], dtype=float)
    d = len(bounds)
    n = max(5, popsize * d)
    pop = np.empty((nbasins, n, d))
    for b in range(nbasins):                                             # Latin hypercube per basin
        seg = (np.arange(n)[:, None] + rng.random((n, d))) / n
        for j in range(d):
            pop[b, :, j] = seg[rng.permutation(n), j]
    clean = lambda e: np.where(np.isfinite(e), e, np.inf)
    active = np.arange(nbasins)
    energies = clean(np.asarray(func(active, lo + pop * (hi - lo)), dtype=float))
    nfev = np.full(nbasins, n)
    nit = np.zeros(nbasins, dtype=int)
    for it in range(1, maxiter + 1):
        fin = np.isfinite(energies).all(axis=1)
        spread = np.std(np.where(np.isfinite(energies), energies, 0.0), axis=1)
        conv = fin & (spread <= atol + tol * np.abs(np.mean(np.where(np.isfinite(energies), energies, 0.0), axis=1)))
        active = np.nonzero(~conv)[0]
        if len(active) == 0:
            break
        k = len(active)
        best = pop[active, np.argmin(energies[active], axis=1)]           # [k, d]
        scale = rng.uniform(mutation[0], mutation[1], size=(k, 1, 1))
        r = np.stack([np.stack([rng.choice(n, 2, replace=False) for _ in range(n)]) for _ in range(k)])   # [k, n, 2]
        pa = np.take_along_axis(pop[active], r[:, :, :1].repeat(d, axis=2), axis=1)
        pb = np.take_along_axis(pop[active], r[:, :, 1:].repeat(d, axis=2), axis=1)
        mutant = best[:, None, :] + scale * (pa - pb)
        cross = rng.random((k, n, d)) < recombination
        jj = rng.integers(0, d, (k, n))
        cross[np.arange(k)[:, None], np.arange(n)[None, :], jj] = True
        trial = np.where(cross, mutant, pop[active])
        out = (trial < 0) | (trial > 1)
        trial[out] = rng.random(int(out.sum()))
        e_trial = clean(np.asarray(func(active, lo + trial * (hi - lo)), dtype=float))
        nfev[active] += n
        nit[active] = it
        better = e_trial <= energies[active]
        pa2, ea2 = pop[active], energies[active]
        pa2[better], ea2[better] = trial[better], e_trial[better]
        pop[active], energies[active] = pa2, ea2
    kbest = np.argmin(energies, axis=1)
    x = lo + pop[np.arange(nbasins), kbest] * (hi - lo)
    return x, energies[np.arange(nbasins), kbest], nfev, nit


def calibrate_all(settings, data, pet, router_function=None, seed=None, popsize=15):
    """Calibrate every requested basin (:256-262).

    All basins search in lock-step: each generation is ONE multi-basin launch of the objective
    (xh_calib_objective_multi), ~3x faster than one launch per basin because a single basin cannot fill the chip.
    Writes the reference's two files per basin (:130-131) and returns {basin: (parameters, kge)}.
    """
    if settings.set_calibrate != 0:
        raise NotImplementedError('set_calibrate = 1 (stream flow) is not supported; see the module docstring')
    basins = expand_str_range(settings.cal_basins)
    cals = [Calibrate(basin_num=b, set_calibrate=0, obs_unit=settings.obs_unit, basin_ids=data.basin_ids,
                      basin_areas=data.area, precip=data.precip, pet=pet, obs=data.cal_obs, tmin=data.tmin,
                      n_months=settings.nmonths, runoff_spinup=settings.runoff_spinup, out_dir=settings.calib_out_dir,
                      device=getattr(settings, 'device', 0)) for b in basins]
    cals = [c for c in cals if c.basin_idx[0].size > 0]
    if not cals:
        return {}
    objs = [c.objective() for c in cals]
    ctx, nosnow = objs[0].ctx, objs[0].nosnow
    npar = objs[0].npar
    obs = np.stack([o.obs for o in objs])

    def func(active, P):
        sel = [objs[i] for i in active]
        return ctx.calib_objective_multi([o.ncell for o in sel], settings.nmonths, settings.runoff_spinup,
                                         P[:, :, :npar], [o.d_pet for o in sel], [o.d_precip for o in sel],
                                         None if nosnow else [o.d_tmin for o in sel],
                                         None if objs[0].d_area is None else [o.d_area for o in sel], obs[active])
    st = time.time()
    try:
        x, ed, nfev, nit = differential_evolution_multi(func, cals[0].bounds, len(cals), popsize=popsize, seed=seed)
    finally:
        for o in objs:
            o.close()
    logging.info('\tCalibrated {} basins in {:.1f} s ({} objective evaluations)'.format(len(cals), time.time() - st,
                                                                                     int(nfev.sum())))
    par_names = 'abcd' + 'm' * (not nosnow)
    results = {}
    for i, c in enumerate(cals):
        c.all_pars[0, :], c.kge_vals[0], c.nfev = x[i], 1 - ed[i], int(nfev[i])
        results[c.basin_num] = (x[i], 1 - ed[i])
        if c.out_dir is not None:
            os.makedirs(c.out_dir, exist_ok=True)
            np.save('{}/kge_result_basin_{}.npy'.format(c.out_dir, c.basin_num), c.kge_vals)
            np.save('{}/{}_parameters_basin_{}.npy'.format(c.out_dir, par_names, c.basin_num), c.all_pars)
    return results
