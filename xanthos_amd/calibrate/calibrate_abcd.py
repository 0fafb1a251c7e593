"""ABCD calibration on MI355X -- mirror of xanthos/calibrate/calibrate_abcd.py.

The reference wraps ``scipy.optimize.differential_evolution`` around ``objective_kge`` (:103-112, :176-213), which
runs ABCD on one basin for ONE parameter vector per call.  Here the objective is evaluated for a whole population
per call by ``xh_calib_objective`` (csrc/xh_calib.hip: members x cells on the GPU, forcing transposed once per basin
and kept in HBM), and the differential evolution itself runs on the device too (csrc/xh_calib_de.hip): ``best1bin``
with SciPy's defaults (popsize 15 x n_parameters members, Latin-hypercube start, dithered mutation in (0.5, 1),
recombination 0.7, tol 0.01, maxiter 1000, the two sampled members distinct from the candidate), generation-synchronous
like SciPy's ``updating='deferred'``, for ALL requested basins at once instead of the reference's serial basin loop
(:256-262).  With several ranks (``launch.current_group()``) the basins are dealt to the ranks largest-first and the
``[n_basins, n_par + 1]`` results gathered on rank 0.  SciPy's driver is unseeded, so the reference's search
trajectory is not reproducible; the objective is the parity target (tests/golden/kge.npz) and the generation step is
checked against a numpy restatement of SciPy's (oracle/de.py).

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
        # the loader's np.nan_to_num of TempMinFile (data_load.py:194-195); precipitation keeps its NaNs (:186)
        self.d_tmin = None if self.nosnow else self.ctx.nan_to_num(tr(tmin))
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


class BasinSet:
    """Several basins prepared for the device: forcing transposed to [month, cell] in HBM, observations, bounds."""

    def __init__(self, cals, n_months, runoff_spinup, obs_unit, device=0):
        self.cals = cals
        self.objs = [c.objective() for c in cals]
        self.ctx = self.objs[0].ctx
        self.nosnow, self.npar = self.objs[0].nosnow, self.objs[0].npar
        self.n_months, self.spinup = int(n_months), int(runoff_spinup)
        self.obs = np.stack([o.obs for o in self.objs])
        self.bounds = cals[0].bounds

    def args(self):
        o = self.objs
        return ([x.ncell for x in o], [x.d_pet for x in o], [x.d_precip for x in o],
                None if self.nosnow else [x.d_tmin for x in o],
                None if o[0].d_area is None else [x.d_area for x in o])

    def evaluate(self, pars):
        """ED for parameter sets pars [nbasins, nmembers, npar] in ONE launch."""
        nc, pet, pr, tn, ar = self.args()
        return self.ctx.calib_objective_multi(nc, self.n_months, self.spinup, np.asarray(pars)[:, :, :self.npar], pet,
                                              pr, tn, ar, self.obs)

    def solver(self, nmembers, seed=0):
        nc, pet, pr, tn, ar = self.args()
        return _hip.CalibDE(self.ctx, nc, self.n_months, self.spinup, nmembers, self.bounds, pet, pr, tn, ar, self.obs,
                            seed=seed, keys=[c.basin_num for c in self.cals])

    def close(self):
        for o in self.objs:
            o.close()


def differential_evolution_device(bset, popsize=15, maxiter=1000, tol=0.01, atol=0.0, mutation=(0.5, 1.0),
                                  recombination=0.7, seed=None, nmembers=None, check_every=4):
    """DE/best/1/bin for every basin of ``bset`` at once, entirely on the device (csrc/xh_calib_de.hip).

    SciPy's defaults as the reference uses them (calibrate_abcd.py:103-112): ``popsize x n_parameters`` members,
    Latin-hypercube start, dither (0.5, 1), recombination 0.7, tol 0.01, maxiter 1000, no polish; selection is
    generation-synchronous (SciPy's ``updating='deferred'``).  The host only enqueues generations, ``check_every`` at
    a time, and reads back how many basins are still searching.  Returns (x [nb, d], fun [nb], nfev [nb], nit [nb]).
    """
    d = len(bset.bounds)
    n = int(nmembers) if nmembers else max(5, popsize * d)
    if seed is None:
        seed = int.from_bytes(os.urandom(8), 'little')       # unseeded like the reference; pass a seed to reproduce
    de = bset.solver(n, seed=seed)
    try:
        de.init()
        done, left = 0, len(bset.cals)
        while done < maxiter and left > 0:
            k = min(check_every, maxiter - done)
            left = de.step(k, tol=tol, atol=atol, mutation=mutation, recombination=recombination)
            done += k
        x, fun, nfev, nit, _ = de.result()
    finally:
        de.close()
    return x, fun, nfev, nit


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
        if polish:
            raise NotImplementedError('polish=True (L-BFGS-B after the search) is not offered; the reference default is False')
        st = time.time()
        bset = BasinSet([self], self.n_months, self.runoff_spinup, self.obs_unit)
        try:
            x, ed, nfev, nit = differential_evolution_device(bset, popsize=popsize, seed=self.seed)
        finally:
            bset.close()
        self._store(x[0], ed[0], int(nfev[0]))
        logging.debug('\t\tFinished calibration for basin {0} which contains {1} grid cells.'.format(
            self.basin_num, self.basin_idx[0].shape[0]))
        logging.debug('\t\tPopulation size:  {}'.format(popsize))
        logging.debug('\t\tParameter values ({}):  {}'.format(','.join(list(self.par_names())), x[0]))
        logging.debug('\t\tKGE:  {}'.format(1 - ed[0]))
        logging.debug('\t\tNumber of function evaluations:  {} in {} generations'.format(int(nfev[0]), int(nit[0])))
        logging.debug('\t\tCalibration time (seconds):  {}'.format(time.time() - st))

    def par_names(self):
        return 'abcd' + 'm' * (not self.nosnow)

    def _store(self, x, ed, nfev, save=True):
        self.all_pars[0, :] = x
        self.kge_vals[0] = 1 - ed
        self.nfev = nfev
        if save and self.out_dir is not None:
            os.makedirs(self.out_dir, exist_ok=True)
            np.save('{}/kge_result_basin_{}.npy'.format(self.out_dir, self.basin_num), self.kge_vals)
            np.save('{}/{}_parameters_basin_{}.npy'.format(self.out_dir, self.par_names(), self.basin_num), self.all_pars)


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


def assign_basins(sizes, n_ranks):
    """Largest-first onto the least-loaded rank (LPT) by ``sizes`` (cells x months). Returns rank of each basin."""
    sizes = np.asarray(sizes, dtype=np.int64)
    load = np.zeros(n_ranks, dtype=np.int64)
    owner = np.empty(len(sizes), dtype=np.int64)
    for b in np.argsort(-sizes, kind='stable'):
        r = int(np.argmin(load))
        owner[b] = r
        load[r] += sizes[b]
    return owner


def gather_results(local, owner, group, root=0):
    """Gather per-basin result rows to ``root``: local [n_local, w] in the rank's basin order -> [n_basins, w].

    ONE collective of ``n_basins x (n_par + 3)`` doubles in all through the job's process group (``group``:
    ``launch.SocketGroup`` or anything with ``rank`` and ``gather``): every rank sends its own rows, the root puts them
    where ``owner`` says."""
    owner = np.asarray(owner)
    got = group.gather(np.ascontiguousarray(local, dtype=np.float64), root=root)
    if group.rank != root:
        return None
    table = np.zeros((len(owner), local.shape[1]))
    for r, rows in enumerate(got):
        table[np.nonzero(owner == r)[0]] = rows
    return table


def _make_calibrate(b, settings, data, pet):
    return Calibrate(basin_num=b, set_calibrate=0, obs_unit=settings.obs_unit, basin_ids=data.basin_ids,
                     basin_areas=data.area, precip=data.precip, pet=pet, obs=data.cal_obs, tmin=data.tmin,
                     n_months=settings.nmonths, runoff_spinup=settings.runoff_spinup, out_dir=settings.calib_out_dir,
                     device=getattr(settings, 'device', 0))


def _calibrate_local(mine, settings, data, pet, seed, popsize, nmembers):
    """This rank's share: rows [len(mine), npar + 3] = (parameters, ED, nfev, nit) and {basin: Calibrate}."""
    npar = 5 if data.tmin is not None else 4
    if not mine:
        return np.zeros((0, npar + 3)), {}
    cals = [_make_calibrate(b, settings, data, pet) for b in mine]
    bset = BasinSet(cals, settings.nmonths, settings.runoff_spinup, settings.obs_unit)
    try:
        x, ed, nfev, nit = differential_evolution_device(bset, popsize=popsize, seed=seed, nmembers=nmembers)
    finally:
        bset.close()
    return np.column_stack([x, ed, nfev, nit]), dict(zip(mine, cals))


def calibrate_all(settings, data, pet, router_function=None, seed=None, popsize=15, nmembers=None, group=None):
    """Calibrate every requested basin (:256-262).

    All basins search in lock-step on the device (differential_evolution_device).  ``group`` = the job's process group
    (``launch.current_group()``; None = one rank): the basins are dealt to the ranks by size (every rank needs the same
    ``seed``), each rank calibrates its share on its own GPU, and rank 0 receives all results in one collective.  Writes the
    reference's two files per basin (:130-131; on rank 0) and returns {basin: (parameters, kge)} (rank 0; {} elsewhere).
    """
    if settings.set_calibrate != 0:
        raise NotImplementedError('set_calibrate = 1 (stream flow) is not supported; see the module docstring')
    basins = expand_str_range(settings.cal_basins)
    basin_ids = np.asarray(data.basin_ids)
    sizes = np.array([(basin_ids == b).sum() for b in basins])
    basins = [b for b, n in zip(basins, sizes) if n > 0]
    sizes = sizes[sizes > 0]
    if not basins:
        return {}
    rank, n_ranks = (group.rank, group.size) if group is not None else (0, 1)
    if n_ranks > 1 and seed is None:
        raise ValueError('a multi-rank calibration needs the same explicit seed on every rank')
    owner = assign_basins(sizes * settings.nmonths, n_ranks)
    mine = [b for b, r in zip(basins, owner) if r == rank]
    st = time.time()
    npar = 5 if data.tmin is not None else 4
    local, cals = _calibrate_local(mine, settings, data, pet, seed, popsize, nmembers)
    table = gather_results(local, owner, group) if n_ranks > 1 else local
    if rank != 0:
        return {}
    logging.info('\tCalibrated {} basins on {} GPU(s) in {:.1f} s ({} objective evaluations)'.format(
        len(basins), n_ranks, time.time() - st, int(table[:, npar + 1].sum())))
    results = {}
    for b, row in zip(basins, table):
        c = cals[b] if b in cals else _make_calibrate(b, settings, data, pet)
        c._store(row[:npar], row[npar], int(row[npar + 1]))
        results[b] = (row[:npar].copy(), 1 - row[npar])
    return results
