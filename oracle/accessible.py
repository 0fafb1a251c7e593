"""Accessible water by basin (oracle; test infrastructure only).

CPU numpy restatement of xanthos/accessible/accessible.py:25-136: yearly runoff totals in km3 per cell (np.sum over
each year's 12 months, NaN propagating, x area / 1e6), basin totals that skip NaN cell-years and cells without a basin,
centred moving mean with averaged end points, the GCAM years, base flow, environmental flow requirement, and
``min(q - efr, baseflow - efr + reservoir capacity)`` clipped at zero.
"""
import numpy as np


def yearly_km3(runoff, area):
    ncell, nmonths = runoff.shape
    ny = nmonths // 12
    q = np.zeros((ncell, ny))
    conv = area / 1e6
    for i in range(ny):
        q[:, i] = np.sum(runoff[:, i * 12:(i + 1) * 12], axis=1) * conv
    return q


def basin_totals(q, basin_ids):
    nb = int(np.max(basin_ids))
    out = np.zeros((nb, q.shape[1]))
    for y in range(q.shape[1]):
        for c in range(q.shape[0]):
            if not np.isnan(q[c, y]) and basin_ids[c] > 0:
                out[basin_ids[c] - 1, y] += q[c, y]
    return out


def rolling_mean_rows(data, window):
    w = np.repeat(1.0, window) / window
    it = int((window - 1) / 2) + 1
    out = np.zeros(data.shape)
    for i in range(data.shape[0]):
        out[i] = np.convolve(data[i], w, 'same')
        out[i, 0] = np.mean(data[i, :it])
        out[i, -1] = np.mean(data[i, data.shape[1] - it:])
    return out


def gcam_years(qs, start_year, end_year, g0, g1, step):
    valid = list(range(start_year, end_year + 1))
    years = list(range(g0, g1 + 1, step))
    return np.stack([qs[:, valid.index(y)] for y in years], axis=1)


def env_flow(map_runoff, pct, start_year, end_year, hist_end_year):
    if start_year > hist_end_year or end_year <= hist_end_year:
        return pct * np.mean(map_runoff, axis=1)
    hey = list(range(start_year, end_year + 1)).index(hist_end_year)
    return pct * np.mean(map_runoff[:, :hey + 1], axis=1)


def accessible(qtot, base, efr, res):
    """accessible.py:121-130.  ``res`` arrives as an [n_basins, 1] column (pandas .values); ``base - efr + res`` then
    broadcasts to a matrix and the minimum over its rows picks the smallest capacity of ANY basin (x + r is monotone in
    r, so the minimum over the column is x + min(res), bit for bit)."""
    rmin = np.min(np.asarray(res, dtype=float))
    ac = np.zeros(qtot.shape)
    for i in range(qtot.shape[1]):
        c = np.minimum(qtot[:, i] - efr, (base[:, i] - efr) + rmin)
        ac[:, i] = np.where(c < 0, 0, c)
    return ac


def accessible_water(runoff, area, basin_ids, bfi, res_capacity, start_year, end_year, hist_end_year, g0, g1, step, window,
                     pct):
    m = basin_totals(yearly_km3(runoff, area), basin_ids)
    qg = gcam_years(rolling_mean_rows(m, window), start_year, end_year, g0, g1, step)
    base = (qg.T * np.asarray(bfi)).T
    efr = env_flow(m, pct, start_year, end_year, hist_end_year)
    return accessible(qg, base, efr, np.asarray(res_capacity)), m
