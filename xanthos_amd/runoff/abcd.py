"""ABCD runoff -- drop-in for xanthos/runoff/abcd.py on MI355X.

Plugin entry point (components.py:236-241):

    abcd_execute(n_basins, basin_ids, pet, precip, tmin, calib_file, n_months, spinup_steps, jobs)
        -> (PET, AET, Q, Sav), each [ncell, n_months]

plus an ``ABCD`` class with the reference's constructor, ``emulate()`` and the ``rsim`` / ``actual_et`` /
``soil_water_storage`` result attributes ([months, cells]) that calibration reads (calibrate_abcd.py:153-159).
Spin-up, per-basin December means and simulation run in csrc/xh_abcd.hip.  ``jobs`` is accepted and ignored: the
joblib basin chunks of abcd_parallel (:357-391) have no counterpart -- every cell is a GPU thread.
"""
import numpy as np

from .. import _hip


def _dense_groups(ids):
    """Map arbitrary basin ids to 0..n-1."""
    uniq, inv = np.unique(np.asarray(ids), return_inverse=True)
    return inv.astype(np.int32), len(uniq)


def _check_spinup(spinup_steps, n_months):
    if spinup_steps < 25:
        # abcd.py:258-266 indexes rows -1, -13, -25 of the spin-up series and re-raises the IndexError
        raise IndexError('Spin-up steps must produce at least 25 months of spin-up; got {}'.format(spinup_steps))
    if spinup_steps > n_months:
        raise IndexError('spin-up ({}) is longer than the series ({})'.format(spinup_steps, n_months))


def abcd_device(ctx, ncell, n_months, spinup_steps, basin_index, n_groups, par_index, d_pars, npar_rows, d_pet,
                d_precip, d_tmin, want=('aet', 'q', 'sav')):
    """Device-resident variant. Returns dict of DeviceArrays for the requested outputs."""
    _check_spinup(spinup_steps, n_months)
    out = {k: ctx.empty((ncell, n_months)) for k in want}
    ctx.abcd(ncell, n_months, spinup_steps, n_groups, basin_index, par_index, npar_rows, d_pars, d_pet, d_precip,
             d_tmin, out.get('aet'), out.get('q'), out.get('sav'))
    return out


def _run(pars_rows, par_index, basin_ids, pet, precip, tmin, n_months, spinup_steps, device=0):
    ctx = _hip.get_context(device)
    ncell = pet.shape[0]
    _check_spinup(spinup_steps, n_months)
    bidx, n_groups = _dense_groups(basin_ids)
    d_pet = ctx.upload(np.asarray(pet)[:, :n_months])
    d_pr = ctx.upload(np.asarray(precip)[:, :n_months])
    d_tn = None if tmin is None else ctx.nan_to_num(ctx.upload(np.asarray(tmin)[:, :n_months]))   # data_load.py:194-195
    pars5 = np.zeros((pars_rows.shape[0], 5))
    pars5[:, :pars_rows.shape[1]] = pars_rows
    d_pars = ctx.upload(pars5)
    res = abcd_device(ctx, ncell, n_months, spinup_steps, bidx, n_groups, par_index, d_pars, pars5.shape[0], d_pet,
                      d_pr, d_tn)
    host = {k: v.download() for k, v in res.items()}
    for b in [d_pet, d_pr, d_tn, d_pars] + list(res.values()):
        if b is not None:
            b.free()
    return host


class ABCD:
    """A hydrology emulator; constructor and results as abcd.ABCD (:18-311), computed on the GPU."""

    def __init__(self, pars, pet, precip, tmin, basin_ids, process_steps, spinup_steps, method='dist'):
        self.nosnow = tmin is None
        self.pars = np.asarray(pars, dtype=np.float64)
        self.basin_ids = np.asarray(basin_ids)
        self.steps = process_steps
        self.spinup_steps = spinup_steps
        self.method = method
        self._pet, self._precip, self._tmin = pet, precip, tmin
        self.pet = np.asarray(pet).T[0:self.steps, :]
        self.precip = np.asarray(precip).T[0:self.steps, :]
        self.tmin = None if self.nosnow else np.asarray(tmin).T[0:self.steps, :]
        self.actual_et = self.rsim = self.soil_water_storage = None

    def emulate(self):
        ncell = self.pars.shape[0]
        host = _run(self.pars, np.arange(ncell, dtype=np.int32), self.basin_ids, self._pet, self._precip, self._tmin,
                    self.steps, self.spinup_steps)
        self.actual_et = host['aet'].T
        self.rsim = host['q'].T
        self.soil_water_storage = host['sav'].T


def abcd_execute(n_basins, basin_ids, pet, precip, tmin, calib_file, n_months, spinup_steps, jobs=-1):
    """Run the ABCD model for every cell. Signature and result of abcd.abcd_execute (:394-422)."""
    prm = calib_file if isinstance(calib_file, np.ndarray) else np.load(calib_file)
    basin_ids = np.asarray(basin_ids)
    min_basin = basin_ids.min()
    # cells outside [min_basin, min_basin + n_basins) are never selected by the reference's chunks (:369-389)
    if basin_ids.max() >= min_basin + n_basins:
        raise ValueError('basin ids exceed n_basins = {}'.format(n_basins))
    host = _run(np.asarray(prm, dtype=np.float64), (basin_ids - 1).astype(np.int32), basin_ids, pet, precip, tmin,
                n_months, spinup_steps)
    return np.array(np.asarray(pet)[:, :n_months]), host['aet'], host['q'], host['sav']
