"""Penman-Monteith monthly PET -- drop-in for xanthos/pet/penman_monteith.py on MI355X.

Same plugin entry point as the reference (components.py:201-202):

    run_pmpet(data, ncells, nlcs, start_yr, end_yr, water_idx, snow_idx, land_cover_years) -> [ncells, nmonths]

``data`` is the DataLoader attribute bag (data_load.py:92-135).  The whole year loop, SetData, et_veg, et_water and
et_snow (penman_monteith.py:17-477) run as one HIP kernel (csrc/xh_pm.hip) through the C-ABI; this module only
moves the arrays.  Unlike the reference it does not mutate ``data`` (nothing downstream reads those fields).
"""
import numpy as np

from .. import _hip

_TABLE_FIELDS = ('cL', 'beta', 'rslimit', 'Tminopen', 'Tminclose', 'VPDclose', 'VPDopen', 'RBLmin', 'RBLmax', 'rc',
                 'emiss', 'alpha', 'lai', 'laimin', 'laimax')


def tables_from(data, nlcs):
    """Host parameter tables (data_load.py:94-117) as the dict Context.pm_pet expects."""
    tab = {}
    for name in _TABLE_FIELDS:
        a = np.asarray(getattr(data, name), dtype=np.float64)
        if a.shape[0] < nlcs:
            raise IndexError('PM table {} has {} rows, nlcs = {}'.format(name, a.shape[0], nlcs))
        tab[name] = np.ascontiguousarray(a[:nlcs])
    return tab


def run_pmpet_device(ctx, tables, ncells, start_yr, end_yr, water_idx, snow_idx, land_cover_years, d_tas, d_tmin,
                     d_rhs, d_wind, d_rsds, d_rlds, d_tairprev, d_lct, d_elev, d_pet=None):
    """Device-resident variant: all d_* are DeviceArrays already in HBM; returns the PET DeviceArray."""
    nmonths = (end_yr - start_yr + 1) * 12
    if d_pet is None:
        d_pet = ctx.empty((ncells, nmonths))
    ctx.pm_pet(tables, ncells, nmonths, start_yr, sorted(land_cover_years), water_idx, snow_idx, d_tas, d_tmin, d_rhs,
               d_wind, d_rsds, d_rlds, d_tairprev, d_lct, d_elev, d_pet)
    return d_pet


def run_pmpet(data, ncells, nlcs, start_yr, end_yr, water_idx, snow_idx, land_cover_years, device=0):
    """Run Penman-Monteith PET on the GPU. Signature and result of penman_monteith.run_pmpet (:394-477)."""
    if nlcs < 7:
        # the reference hard-codes albedo rows 0 and 6 (:361, :377) and fails with IndexError below 7 classes
        raise IndexError('index 6 is out of bounds for axis 0 with size {}'.format(nlcs))
    ctx = _hip.get_context(device)
    nmonths = (end_yr - start_yr + 1) * 12
    up = lambda a: ctx.nan_to_num(ctx.upload(np.asarray(a)[:, :nmonths]))      # loader transform (data_load.py:120-125)
    lct = np.asarray(data.lct_load, dtype=np.float64)
    if lct.shape[1] != nlcs or lct.shape[2] != len(land_cover_years):
        raise ValueError('lct_load must be [ncell, nlcs, n land-cover years]; got {}'.format(lct.shape))
    elev = np.asarray(data.elev, dtype=np.float64).reshape(-1)
    if elev.size != ncells:
        raise ValueError('elev must have one value per cell')
    bufs = [up(data.tair_load), up(data.TMIN_load), up(data.rhs_load), up(data.wind_load), up(data.rsds_load),
            up(data.rlds_load), up(data.tairprev_load), ctx.upload(lct), ctx.upload(elev)]
    d_pet = run_pmpet_device(ctx, tables_from(data, nlcs), ncells, start_yr, end_yr, water_idx, snow_idx,
                             land_cover_years, *bufs)
    out = d_pet.download()
    for b in bufs + [d_pet]:
        b.free()
    return out
