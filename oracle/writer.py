"""Output aggregation (oracle; test infrastructure only).

numpy restatement of the array math of xanthos/data_writer/out_writer.py: agg_to_year (:237-248), the mm -> km3
conversion in write() (:111-112, rows x area / 1e6) and agg_spatial (:250-265).  pandas' groupby sum skips NaN
(an all-NaN block sums to 0), mean skips NaN (all-NaN -> NaN); ids without cells give NaN rows (left merge, :261).
"""
import warnings

import numpy as np


def agg_to_year(arr, func='sum'):
    a = np.asarray(arr, dtype=float).reshape(arr.shape[0], -1, 12)
    if func == 'sum':
        return np.nansum(a, axis=2)
    with warnings.catch_warnings():
        warnings.simplefilter('ignore', RuntimeWarning)
        return np.nanmean(a, axis=2)


def mm_to_km3(arr, grid_areas):
    return np.asarray(arr, dtype=float) * (np.asarray(grid_areas, dtype=float) / 1e6)[:, None]


def agg_spatial(arr, id_map, n_ids, first_id=1):
    """Rows = ids first_id .. first_id + n_ids - 1 (the reference's names table), NaN where an id has no cells."""
    arr = np.asarray(arr, dtype=float)
    out = np.full((n_ids, arr.shape[1]), np.nan)
    for k in range(n_ids):
        sel = np.asarray(id_map) == k + first_id
        if sel.any():
            out[k] = np.nansum(arr[sel], axis=0)
    return out
