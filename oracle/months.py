"""Month tables (oracle; test infrastructure only).

Restates xanthos/utils/general.py:15-50 (``set_month_arrays``): rows of
``[year, month_index, days_in_month]`` where a year is leap iff ``year % 4 == 0``
(general.py:37) -- NOT the Gregorian rule.  Penman-Monteith uses
``calendar.isleap`` instead (penman_monteith.py:57); both are kept.
"""
import calendar

import numpy as np

_DAYS = (31, 28, 31, 30, 31, 30, 31, 31, 30, 31, 30, 31)


def set_month_arrays(n_months, start_year, end_year):
    """[n_months, 3] int table; leap rule year % 4 == 0 (general.py:15-50)."""
    tab = np.zeros((n_months, 3), dtype=int)
    row = 0
    for year in range(start_year, end_year + 1):
        for mth in range(12):
            ndays = _DAYS[mth] + (1 if (mth == 1 and year % 4 == 0) else 0)
            tab[row] = (year, mth, ndays)
            row += 1
    return tab


def pm_days_in_month(year):
    """Days per month as Penman-Monteith sees them (penman_monteith.py:57-60)."""
    d = np.array(_DAYS, dtype=np.int64)
    if calendar.isleap(year):
        d[1] = 29
    return d


def pm_land_cover_index(year, land_cover_years):
    """Which land-cover slice a simulation year uses (penman_monteith.py:33-43)."""
    lc = sorted(land_cover_years)
    if year >= lc[-1]:
        return len(lc) - 1
    for i, x in enumerate(lc):
        if x - year >= -4:
            return i
    return len(lc) - 1
