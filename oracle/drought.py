"""Drought statistics (oracle; test infrastructure only).

CPU numpy restatement of xanthos/drought/drought_stats.py:
  getthresh ............... reshape to (years, periods, cells), 10th percentile over the years      :150-171
  calculate_thresholds .... the reference-period slice [smonth:emonth] (emonth is a LENGTH, sic)    :69-83
  droughtstats ............ duration / severity / intensity recurrences over time                  :85-148

Third-party arithmetic: ``np.percentile`` (numpy >= 1.22 "linear" method: virtual index (n-1) q, _lerp with the
t >= 0.5 branch); the oracle calls numpy itself, the HIP kernel restates the algorithm (csrc/xh_drought.hip).
Arrays are [ntime, ngrid] like the reference's.
"""
import numpy as np

MONTHS_IN_YEAR = 12


def getthresh(histout, nper, quantile=0.1):
    ntime, ngrid = histout.shape
    nyear = int(ntime / nper)
    return np.percentile(np.reshape(histout, (nyear, nper, ngrid)), quantile * 100, axis=0)


def calculate_thresholds(histout, start_year, threshold_start_year, threshold_end_year, nper):
    smonth = (threshold_start_year - start_year) * MONTHS_IN_YEAR
    emonth = (threshold_end_year + 1 - threshold_start_year) * MONTHS_IN_YEAR
    return getthresh(histout[smonth:emonth, :], nper)


def droughtstats(hydroout, threshvals):
    ntime = hydroout.shape[0]
    nthresh = threshvals.shape[0]
    S = np.zeros_like(hydroout)
    I = np.zeros_like(hydroout)
    D = np.zeros_like(hydroout)
    dry = hydroout[0] < threshvals[0]
    D[0] = np.where(dry, 1.0, 0.0)
    with np.errstate(divide='ignore', invalid='ignore'):
        S[0] = I[0] = np.where(dry, (threshvals[0] - hydroout[0]) / threshvals[0], 0.0)
        for t in range(1, ntime):
            th = threshvals[t % nthresh]
            dry = hydroout[t] < th
            D[t] = np.where(dry, D[t - 1] + 1, 0.0)
            S[t] = np.where(dry, S[t - 1] + (th - hydroout[t]) / th, 0.0)
            I[t] = np.where(dry, S[t] / D[t], 0.0)
    return S, I, D
