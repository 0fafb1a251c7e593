"""ABCD(+snow) monthly water balance (oracle; test infrastructure only).

CPU numpy restatement of xanthos/runoff/abcd.py.  State is carried month to
month as three vectors (snowpack, soil moisture, groundwater) instead of the
reference's full ``[months, cells]`` work arrays; the arithmetic per month is
the reference's ``abcd_dist`` (:171-228) in the same order.

Reference map:
  parameters / thresholds ............... ABCD.__init__        :41-100
  rain / snow split ..................... set_rain_and_snow    :141-169
  one month ............................. abcd_dist            :171-228
  spin-up -> per-basin December means ... spinup, set_vals     :246-294
  simulation ............................ simulate, emulate    :296-311
  basin chunks + joblib threads ......... _run_basins, abcd_parallel, abcd_execute :314-422
"""
import numpy as np

TRAIN = 2.5   # :99
TSNOW = 0.6   # :100
SM_INIT = 100.0   # :82-83  inv[1]
GW_INIT = 500.0   # :84     inv[2]


def _split_rain_snow(p, tmin):
    """set_rain_and_snow (:141-169): p, tmin are [months, cells]."""
    if tmin is None:
        return p, None
    rain = np.zeros_like(p)
    snow = np.zeros_like(p)
    allrain = tmin > TRAIN
    mixed = (tmin <= TRAIN) & (tmin >= TSNOW)
    allsnow = tmin < TSNOW
    snow[mixed] = p[mixed] * (TRAIN - tmin[mixed]) / (TRAIN - TSNOW)
    rain[allrain] = p[allrain]
    rain[mixed] = p[mixed] - snow[mixed]
    snow[allsnow] = p[allsnow]
    return rain, snow


def _march(a, b, c, d, m, pet, precip, tmin, sm0, gw0, steps, keep=None):
    """Run ``steps`` months of abcd_dist (:171-228) from soil moisture ``sm0`` / groundwater ``gw0``.

    pet/precip/tmin: [months, cells].  Returns (aet, q, sav, sm_keep, gw_keep) where the first three are
    [steps, cells] and the *_keep lists hold soil-moisture / groundwater rows for the month indices in ``keep``.
    """
    ncell = pet.shape[1]
    rain, snow = _split_rain_snow(precip[:steps], None if tmin is None else tmin[:steps])
    a2 = a * 2
    b_over_a = b / a
    d1 = d + 1

    aet = np.empty((steps, ncell))
    q = np.empty((steps, ncell))
    sav = np.empty((steps, ncell))
    gw_rows = {}
    snowpack = np.zeros(ncell)          # SN0 = 0 (:98)
    sm_prev = sm0
    gw_prev = gw0
    with np.errstate(invalid='ignore', over='ignore'):
        for i in range(steps):
            snm = np.zeros(ncell)
            if tmin is not None:
                t = tmin[i]
                snowpack = snowpack + snow[i]
                allrain = t > TRAIN
                mixed = (t <= TRAIN) & (t >= TSNOW)
                snm[allrain] = snowpack[allrain] * m[allrain]
                snm[mixed] = (snowpack[mixed] * m[mixed]) * ((TRAIN - t[mixed]) / (TRAIN - TSNOW))
                snowpack = snowpack - snm
            if i == 0:
                w = rain[i] + sm_prev                   # no snow-melt in the first month (:200-201)
            else:
                w = rain[i] + sm_prev + snm
            rpt = (w + b) / a2
            y = rpt - np.sqrt(np.square(rpt) - (w * b_over_a))
            sm = y * np.exp(-pet[i] / b)
            awet = w - y
            c_awet = c * awet
            gw = (gw_prev + c_awet) / d1
            e = np.minimum(pet[i], np.maximum(0, y - sm))
            sm = y - e
            aet[i] = e
            sav[i] = sm
            q[i] = (awet - c_awet) + d * gw
            if keep is not None and (i in keep):
                gw_rows[i] = gw
            sm_prev, gw_prev = sm, gw
    if keep is None:
        return aet, q, sav
    return aet, q, sav, [sav[k] for k in keep], [gw_rows[k] for k in keep]


def basin_initial_state(sm_dec, gw_dec, basin_ids):
    """set_vals (:246-282): per-basin mean over the three Decembers of the per-row nan-mean over cells."""
    import warnings
    sm_dec = np.asarray(sm_dec)
    gw_dec = np.asarray(gw_dec)
    sm0 = np.empty(basin_ids.shape)
    gw0 = np.empty(basin_ids.shape)
    with warnings.catch_warnings():
        warnings.simplefilter('ignore', RuntimeWarning)
        for bid in np.unique(basin_ids):
            sel = basin_ids == bid
            sm0[sel] = np.mean(np.nanmean(sm_dec[:, sel], axis=1))
            gw0[sel] = np.mean(np.nanmean(gw_dec[:, sel], axis=1))
    return sm0, gw0


class ABCD:
    """Same constructor, ``emulate()`` and result attributes as abcd.ABCD (:18-311)."""

    def __init__(self, pars, pet, precip, tmin, basin_ids, process_steps, spinup_steps, method='dist'):
        self.nosnow = tmin is None
        self.a = pars[:, 0]
        self.b = pars[:, 1] * 1000
        self.c = pars[:, 2]
        self.d = pars[:, 3]
        self.m = pars[:, 4] if not self.nosnow else np.zeros(pars.shape[0])
        self.basin_ids = basin_ids
        self.steps = process_steps
        self.spinup_steps = spinup_steps
        self.pet = pet.T[0:self.steps, :]
        self.precip = precip.T[0:self.steps, :]
        self.tmin = None if self.nosnow else tmin.T[0:self.steps, :]
        self.actual_et = self.rsim = self.soil_water_storage = None
        self.sm0 = self.gw0 = None

    def emulate(self):
        if self.spinup_steps < 25:
            # the reference indexes rows -1, -13, -25 of the spin-up series (:258-266)
            raise IndexError('Spin-up steps must produce at least 25 months; got {}'.format(self.spinup_steps))
        n = self.pet.shape[1]
        s = self.spinup_steps
        keep = [s - 1, s - 13, s - 25]
        _, _, _, sm_dec, gw_dec = _march(self.a, self.b, self.c, self.d, self.m, self.pet, self.precip,
                                         self.tmin, np.full(n, SM_INIT), np.full(n, GW_INIT), s, keep=keep)
        self.sm0, self.gw0 = basin_initial_state(sm_dec, gw_dec, self.basin_ids)
        self.actual_et, self.rsim, self.soil_water_storage = _march(
            self.a, self.b, self.c, self.d, self.m, self.pet, self.precip, self.tmin,
            self.sm0, self.gw0, self.steps)


def _run_basins(basin_nums, pars_abcdm, basin_ids, pet, precip, tmin, n_months, spinup_steps):
    """_run_basins (:314-354)."""
    idx = np.where(np.isin(basin_ids, basin_nums))
    pars = pars_abcdm[basin_ids - 1][idx]
    he = ABCD(pars, pet[idx], precip[idx], None if tmin is None else tmin[idx],
              basin_ids[idx], n_months, spinup_steps)
    he.emulate()
    return idx, he.actual_et.T, he.rsim.T, he.soil_water_storage.T


def abcd_parallel(n_basins, pars, basin_ids, pet, precip, tmin, n_months, spinup_steps, jobs=-1):
    """abcd_parallel (:357-391): basin chunks on joblib threads. Returns (aet, q, sav), each [ncell, n_months]."""
    n_chunks = 8 if jobs < 1 else jobs * 2
    min_basin = min(basin_ids)
    chunks = np.array_split(np.arange(min_basin, min_basin + n_basins), n_chunks)
    chunks = [c for c in chunks if len(c)]
    args = (pars, basin_ids, pet, precip, tmin, n_months, spinup_steps)
    try:
        from joblib import Parallel, delayed
        res = Parallel(n_jobs=jobs, backend='threading')(delayed(_run_basins)(c, *args) for c in chunks)
    except ImportError:
        res = [_run_basins(c, *args) for c in chunks]
    aet = np.empty((len(basin_ids), n_months))
    q = np.empty_like(aet)
    sav = np.empty_like(aet)
    for idx, e, r, s in res:
        aet[idx], q[idx], sav[idx] = e, r, s
    return aet, q, sav


def abcd_execute(n_basins, basin_ids, pet, precip, tmin, calib_file, n_months, spinup_steps, jobs):
    """abcd_execute (:394-422); ``calib_file`` may be a path or an ndarray [n_basins, 5]."""
    prm = calib_file if isinstance(calib_file, np.ndarray) else np.load(calib_file)
    aet, q, sav = abcd_parallel(n_basins, prm, basin_ids, pet, precip, tmin, n_months, spinup_steps, jobs)
    return pet[:, :n_months].copy(), aet, q, sav
