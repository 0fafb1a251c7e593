"""ABCD calibration objective (oracle; test infrastructure only).

CPU numpy restatement of xanthos/calibrate/calibrate_abcd.py:134-213: run ABCD on one basin with one
parameter vector spread over the basin's cells, aggregate the simulated runoff to a monthly basin series and
score it against observations with the Kling-Gupta distance ``ED = 1 - KGE``.

The differential-evolution search that wraps this objective (calibrate_abcd.py:103-112) is
``scipy.optimize.differential_evolution`` with no seed: its trajectory is not reproducible, so only the
objective is a parity target ("parity unpinned" for the optimiser itself).
"""
import numpy as np

from .abcd import ABCD


def basin_runoff(pars, set_calibrate, pet, precip, tmin, n_months, runoff_spinup,
                 obs_unit, bsn_areas, basin_idx=None, arr_shp=None, routing_func=None):
    """Monthly basin series for one parameter vector (calibrate_abcd.py:134-173)."""
    ncell = pet.shape[0]
    pars = np.repeat(np.asarray(pars, dtype=float)[np.newaxis, :], ncell, axis=0)
    he = ABCD(pars, pet, precip, tmin, np.zeros(ncell), n_months, runoff_spinup)
    he.emulate()
    if set_calibrate == 0:
        if obs_unit == 'km3_per_mth':
            return np.nansum(he.rsim * bsn_areas * 1e-6, 1)
        if obs_unit == 'mm_per_mth':
            return np.nansum(he.rsim, 1)
        raise ValueError(obs_unit)
    rsim = np.zeros(shape=arr_shp)
    np.put(rsim, basin_idx, he.rsim)          # sic: flat put, as the reference does (:170)
    return routing_func(rsim)


def kge_distance(modelled, observed):
    """ED = sqrt((r-1)^2 + (alpha-1)^2 + (beta-1)^2) (calibrate_abcd.py:196-213)."""
    relvar = np.std(modelled) / np.std(observed)
    bias = np.mean(modelled) / np.mean(observed)
    corr = np.corrcoef(observed, modelled)[1, 0]
    return (((corr - 1) ** 2) + ((relvar - 1) ** 2) + ((bias - 1) ** 2)) ** 0.5


def objective_kge(pars, set_calibrate, pet, precip, tmin, n_months, runoff_spinup, obs_unit,
                  bsn_areas, bsn_robs, basin_idx=None, arr_shp=None, routing_func=None):
    """objective_kge (calibrate_abcd.py:176-213) with basin_runoff as the model function."""
    modelled = basin_runoff(pars, set_calibrate, pet, precip, tmin, n_months, runoff_spinup,
                            obs_unit, bsn_areas, basin_idx, arr_shp, routing_func)
    return kge_distance(modelled, bsn_robs)
