"""Accessible water by basin (mirror of xanthos/accessible/accessible.py:25-152).

``AccessibleWater(settings, ref, runoff)`` with the reference's settings fields (ResCapacityFile, BfiFile, HistEndYear,
GCAM_StartYear / EndYear / YearStep, MovingMeanWindow, Env_FlowPercent) and output file
(``accessible_water_km3peryr_<name>.csv``).  The grid-sized work -- yearly totals of the 12 monthly values of every
cell in numpy's summation order, mm -> km3, and the sums over the cells of each basin -- runs on the GPU
(``xh_agg_time`` mode 2, ``xh_agg_spatial``) on the runoff array the pipeline left in HBM; what remains is
arithmetic on a [basins, years] table, done here with numpy.
"""
import logging
import os

import numpy as np

from .. import _hip


def basin_year_totals(ctx, runoff, area, basin_ids):
    """[n_basins, n_years] km3 per year: accessible.py:37-54.  ``runoff`` [ncell, nmonths], host or DeviceArray."""
    src = runoff if isinstance(runoff, _hip.DeviceArray) else ctx.upload(np.ascontiguousarray(runoff, dtype=np.float64))
    ncell, nmonths = src.shape
    ny = int(nmonths / 12)
    d_scale = ctx.upload(np.asarray(area, dtype=np.float64) / 1e6)
    d_year = ctx.empty((ncell, ny))
    if ny * 12 != nmonths:
        raise ValueError("runoff must cover whole years ({} months)".format(nmonths))
    ctx.agg_time(ncell, nmonths, 12, 2, d_scale, src, d_year)
    ids = np.asarray(basin_ids).reshape(-1).astype(np.int64)
    nb = int(ids.max())
    group = np.where(ids > 0, ids - 1, -1).astype(np.int32)
    d_out = ctx.empty((nb, ny))
    ctx.agg_spatial(ncell, ny, nb, group, d_year, d_out)
    out = np.nan_to_num(d_out.download(), nan=0.0)        # a basin without cells stays at the zero it started from
    for a in (d_scale, d_year, d_out) + (() if src is runoff else (src,)):      # a caller's array stays resident
        a.free()
    return out


def rolling_window_filter(data, window):
    """Centred moving mean along each row with averaged end points (accessible.py:82-106, Dimension = 0)."""
    weights = np.repeat(1.0, window) / window
    half = int((window - 1) / 2) + 1
    out = np.zeros(data.shape, dtype=float)
    n = data.shape[1]
    for i in range(data.shape[0]):
        out[i, :] = np.convolve(data[i, :], weights, 'same')
        out[i, 0] = np.mean(data[i, :half])
        out[i, n - 1] = np.mean(data[i, n - half:])
    return out


def q_in_gcam_years(qs, settings):
    valid = list(range(settings.StartYear, settings.EndYear + 1))
    years = list(range(settings.GCAM_StartYear, settings.GCAM_EndYear + 1, settings.GCAM_YearStep))
    out = np.zeros((qs.shape[0], len(years)), dtype=float)
    for k, y in enumerate(years):
        out[:, k] = qs[:, valid.index(y)]
    return out


def accessible_water(qtot, base, efr, res):
    """min(q - efr, baseflow - efr + reservoir capacity), clipped at zero (accessible.py:121-130).

    ``res`` is the [n_basins, 1] column the reference reads with pandas; added to a [n_basins] vector it broadcasts
    to a matrix and the column-wise minimum then takes the SMALLEST capacity of any basin for every basin.  Kept as is.
    """
    res = np.asarray(res, dtype=float)
    ac = np.zeros(qtot.shape, dtype=float)
    for i in range(qtot.shape[1]):
        a = qtot[:, i] - efr
        b = base[:, i] - efr + res
        c = np.min(np.vstack((a, b)), axis=0)
        ac[:, i] = np.where(c < 0, 0, c)
    return ac


def gen_gcam_output(filename, data, names, settings):
    years = [str(y) for y in range(settings.GCAM_StartYear, settings.GCAM_EndYear + 1, settings.GCAM_YearStep)]
    text = np.asarray(data, dtype=float).astype(str)      # the reference formats through ndarray.astype(str) (:148)
    with open(filename, 'w') as fh:
        fh.write('id,name,' + ','.join(years) + '\n')
        for k in range(len(names)):
            fh.write('{},{},{}\n'.format(k + 1, names[k], ','.join(text[k])))


def AccessibleWater(settings, ref, runoff):
    """Accessible water per basin and GCAM year; returns the [n_basins, n_gcam_years] table it writes."""
    ctx = _hip.get_context(getattr(settings, 'device', 0))
    import pandas as pd        # the two basin tables are read as the reference reads them (pandas' float parser)
    res = pd.read_csv(settings.ResCapacityFile, header=None, names=['res_capacity']).values
    bfi = pd.read_csv(settings.BfiFile)['bfi_avg'].values

    map_runoff = basin_year_totals(ctx, runoff, ref.area, ref.basin_ids)
    qs = rolling_window_filter(map_runoff, settings.MovingMeanWindow)
    q_gcam = q_in_gcam_years(qs, settings)
    bflow = np.transpose(np.transpose(q_gcam) * np.array(bfi))

    if settings.StartYear > settings.HistEndYear:
        logging.warning('No historical data used in calculating Environmental Flow '
                        'Requirements (EFR) per basin for Accessible Water')
        efr = settings.Env_FlowPercent * np.mean(map_runoff, axis=1)
    elif settings.EndYear <= settings.HistEndYear:
        efr = settings.Env_FlowPercent * np.mean(map_runoff, axis=1)
    else:
        hey = list(range(settings.StartYear, settings.EndYear + 1)).index(settings.HistEndYear)
        efr = settings.Env_FlowPercent * np.mean(map_runoff[:, :(hey + 1)], axis=1)

    ac = accessible_water(q_gcam, bflow, efr, res)
    os.makedirs(settings.OutputFolder, exist_ok=True)
    filename = os.path.join(settings.OutputFolder, 'accessible_water_km3peryr_{}.csv'.format(settings.OutputNameStr))
    gen_gcam_output(filename, ac, ref.basin_names, settings)
    return ac
