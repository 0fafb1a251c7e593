"""BASELINE config 5 as a device-resident workload: ABCD differential-evolution calibration of all 235 basins of the
synthetic 67,420-cell world with a 512-member population, ``set_calibrate = 0``, ``obs_unit = km3_per_mth``,
480 + 120 months.

Everything is produced on the device: the forcing (xh_synth_forcing), PET (the Penman-Monteith kernel on that
forcing), each basin's ``[month, cell]`` blocks (row gather + transpose), and the "observed" runoff = the basin series
the objective kernel gives at the world's hidden true ABCD parameters, times N(1, 0.05) noise.  ``bench.py --workload
calib`` and tests/test_gpu_calib.py drive this class; with ``basins`` it holds one rank's share of a multi-GPU run.
"""
import numpy as np

from .. import _hip, synth
from ..pet import penman_monteith as pm_mod
from .calibrate_abcd import LB, UB

BOUNDS = [(LB, UB), (LB, 8 - LB), (LB, UB), (LB, UB), (LB, UB)]          # calibrate_abcd.py:62-64


class Config5:
    def __init__(self, ctx, nmembers=512, nmonths=480, spinup=120, seed=1, start_year=1971, world=None, basins=None,
                 noise=0.05):
        self.ctx, self.nmonths, self.spinup, self.nmembers = ctx, int(nmonths), int(spinup), int(nmembers)
        w = self.world = world if world is not None else synth.make_world()
        nm = self.nmonths
        f = {k: ctx.empty((w.ncell, nm)) for k in synth.FORCING_NAMES}
        d_lat = ctx.upload(w.latitude)
        ctx.synth_forcing(synth.MASTER_SEED + 5, w.ncell, nm, d_lat, f, nan_frac=0.0)
        # tairprev = previous cell's temperature, zeros for cell 0 (data_load.py:128-129)
        d_prev = ctx.empty((w.ncell, nm)).zero()
        ctx._check(_hip.lib().xh_memcpy_d2d(ctx.handle, d_prev.ptr + nm * 8, f['tas'].ptr, (w.ncell - 1) * nm * 8))
        d_lct, d_elev = ctx.upload(w.lct), ctx.upload(np.asarray(w.elev, dtype=np.float64).reshape(-1))
        d_pet = pm_mod.run_pmpet_device(ctx, pm_mod.tables_from(w, w.nlcs), w.ncell, start_year,
                                        start_year + nm // 12 - 1, 0, 6, w.lc_years, f['tas'], f['tmin'], f['rhs'],
                                        f['wind'], f['rsds'], f['rlds'], d_prev, d_lct, d_elev)
        order = np.argsort(w.basin_ids, kind='stable')
        counts_all = np.bincount(w.basin_ids, minlength=w.n_basins + 1)[1:]
        start = np.concatenate([[0], np.cumsum(counts_all)])
        self.basins = list(range(1, w.n_basins + 1)) if basins is None else [int(b) for b in basins]
        self.keys = list(self.basins)
        self.counts = np.array([counts_all[b - 1] for b in self.basins], dtype=np.int64)
        self.cells = [order[start[b - 1]:start[b]] for b in self.basins]
        self.blocks = []
        for cells in self.cells:
            n = len(cells)
            d_rows = ctx.upload(cells, dtype=np.int64)
            blk = {}
            for name, src in (('pet', d_pet), ('precip', f['precip']), ('tmin', f['abcd_tmin'])):
                tmp = ctx.empty((n, nm))
                ctx.gather_rows(src, d_rows, n, nm, tmp)
                blk[name] = ctx.empty((nm, n))
                ctx.transpose(tmp, n, nm, blk[name])
                tmp.free()
            blk['area'] = ctx.upload(w.area[cells])
            d_rows.free()
            self.blocks.append(blk)
        ctx.sync()
        for b in list(f.values()) + [d_lat, d_prev, d_lct, d_elev, d_pet]:
            b.free()
        # observations: basin series at the true parameters x N(1, noise)
        nb = len(self.basins)
        true = np.stack([w.abcd_pars[b - 1] for b in self.basins])[:, None, :]
        _, series = ctx.calib_objective_multi(self.counts, nm, self.spinup, true, self._lst('pet'), self._lst('precip'),
                                              self._lst('tmin'), self._lst('area'), np.ones((nb, nm)),
                                              want_series=True)
        idx = (np.asarray(self.basins, dtype=np.uint64)[:, None] * np.uint64(4096) + np.arange(nm, dtype=np.uint64)[None, :])
        self.obs = series[:, 0, :] * (1.0 + noise * synth.normal(synth.MASTER_SEED + 5, 60, idx))
        self.de = _hip.CalibDE(ctx, self.counts, nm, self.spinup, self.nmembers, BOUNDS, self._lst('pet'),
                               self._lst('precip'), self._lst('tmin'), self._lst('area'), self.obs, seed=seed,
                               keys=self.keys)

    def _lst(self, name):
        return [blk[name] for blk in self.blocks]

    @property
    def member_cell_months(self):
        """Member-cell-months one generation simulates (spin-up months included)."""
        return int(self.nmembers * self.counts.sum() * (self.nmonths + self.spinup))

    def host_basin(self, b):
        """Host copies of basin index ``b``'s forcing in the reference's [cell, month] layout, and its areas."""
        blk = self.blocks[b]
        return {'pet': blk['pet'].download().T.copy(), 'precip': blk['precip'].download().T.copy(),
                'tmin': blk['tmin'].download().T.copy(), 'area': blk['area'].download()}

    def evaluate_one(self, b, pars):
        """ED of parameter sets ``pars`` [n, 5] for basin index ``b`` alone."""
        blk = self.blocks[b]
        return self.ctx.calib_objective(int(self.counts[b]), self.nmonths, self.spinup, pars, blk['pet'], blk['precip'],
                                        blk['tmin'], blk['area'], self.obs[b])

    def close(self):
        self.de.close()
        for blk in self.blocks:
            for a in blk.values():
                a.free()
        self.blocks = []
