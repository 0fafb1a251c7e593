"""Month tables for the harness (host logic).

``set_month_arrays`` mirrors xanthos/utils/general.py:15-50: rows of [year, month_index, days] where a year is
leap iff ``year % 4 == 0`` (general.py:37 -- not the Gregorian rule; the routing sub-step count depends on it,
components.py:276,288).  Penman-Monteith's own calendar (calendar.isleap, penman_monteith.py:57) is applied inside
the PM kernel.
"""
import numpy as np

_DAYS = np.array([31, 28, 31, 30, 31, 30, 31, 31, 30, 31, 30, 31])


def set_month_arrays(n_months, start_year, end_year):
    years = np.repeat(np.arange(start_year, end_year + 1), 12)
    mths = np.tile(np.arange(12), end_year - start_year + 1)
    days = _DAYS[mths] + ((mths == 1) & (years % 4 == 0))
    tab = np.stack([years, mths, days], axis=1).astype(int)
    if tab.shape[0] != n_months:
        raise ValueError('n_months = {} does not match {}..{}'.format(n_months, start_year, end_year))
    return tab
