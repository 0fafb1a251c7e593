"""Device-resident PET -> runoff -> routing pipeline.

One object owns the static grid data and the six output arrays in HBM and enqueues the three stages back to back
on the context's stream; nothing crosses PCIe between stages.  ``components.Components`` (the reference-shaped
harness), ``bench.py`` and the multi-GPU sharding all drive this class; the per-stage plugin functions in
``pet/``, ``runoff/`` and ``routing/`` are the host-array entry points around the same C-ABI calls.
"""
import os

import numpy as np

from . import _hip
from .pet import penman_monteith as pm_mod
from .routing import mrtm as mrtm_mod
from .utils import set_month_arrays

FEED_DEFAULT = '1'      # measured on MI355X: 26.4 -> 25.35 ms per full-grid step (profiles/round4/feed_first_block.txt)
FORCING = ('tas', 'tmin', 'rhs', 'wind', 'rsds', 'rlds', 'precip', 'abcd_tmin')
OUTPUTS = ('pet', 'aet', 'q', 'sav', 'chs', 'avg')


def file_range_of(arr):
    """(path, byte offset) of the first element of a C-contiguous float64 np.memmap in its file, or None.  The position
    is taken from the addresses (the array's data pointer against the start of its mapping), not from ``arr.offset``,
    which a slice of a memory map inherits unchanged from its parent."""
    import mmap
    if not (isinstance(arr, np.memmap) and arr.dtype == np.float64 and arr.dtype.isnative and arr.flags.c_contiguous
            and getattr(arr, 'filename', None) is not None and getattr(arr, '_mmap', None) is not None):
        return None
    base = np.frombuffer(arr._mmap, dtype=np.uint8)
    delta = arr.ctypes.data - base.ctypes.data
    if delta < 0 or delta + arr.nbytes > base.size:
        return None
    map_start = arr.offset - arr.offset % mmap.ALLOCATIONGRANULARITY        # where numpy placed the mapping in the file
    return str(arr.filename), map_start + delta


class DevicePipeline:
    """PM -> ABCD -> MRTM for one set of cells (the whole grid, or one rank's shard) on one GPU."""

    def __init__(self, ctx, *, ncell, nmonths, start_year, basin_ids, abcd_pars, pm_tables, lct, elev, lc_years,
                 um, flow_dist, velocity, area, abcd_spinup, routing_spinup, water_idx=0, snow_idx=6, use_snow=True,
                 route_flags=0, chs_prev=None, plan_async=False):
        self.ctx = ctx
        self.ncell, self.nmonths, self.start_year = int(ncell), int(nmonths), int(start_year)
        self.end_year = self.start_year + self.nmonths // 12 - 1
        self.abcd_spinup, self.routing_spinup = int(abcd_spinup), int(routing_spinup)
        self.water_idx, self.snow_idx, self.use_snow = water_idx, snow_idx, use_snow
        self.lc_years = sorted(lc_years)
        self.pm_tables = pm_tables
        self.route_flags = route_flags
        basin_ids = np.asarray(basin_ids)
        uniq, inv = np.unique(basin_ids, return_inverse=True)
        self.basin_index = inv.astype(np.int32)
        self.n_groups = len(uniq)
        self.par_index = (basin_ids - 1).astype(np.int32)          # row of abcd_pars = basin id - 1 (abcd.py:332)
        self.npar_rows = int(np.asarray(abcd_pars).shape[0])
        up = ctx.upload
        self.d_pars = up(np.asarray(abcd_pars, dtype=np.float64))
        self.d_lct = up(lct)
        self.d_elev = up(np.asarray(elev, dtype=np.float64).reshape(-1))
        self.um = um
        self.d_flow_dist, self.d_velocity, self.d_area = up(flow_dist), up(velocity), up(area)
        # initial channel storage (future mode, data_load.py:427-438); None = zeros
        self.d_S0 = up(chs_prev) if chs_prev is not None and np.any(np.asarray(chs_prev) != 0) else None
        self.ndays = set_month_arrays(self.nmonths, self.start_year, self.end_year)[:, 2]
        self.forcing = {}
        self.d_tairprev = None
        self.out = {k: ctx.empty((self.ncell, self.nmonths)) for k in OUTPUTS}
        # The routing plan (partition of the networks, 50-70 ms of host time at the full grid) touches neither the
        # context's stream nor the arrays above; with plan_async it is made on a host thread while the caller uploads the
        # forcing (run_model()), and `plan` waits for it.
        self._plan, self._plan_thread, self._plan_error = None, None, None
        if um is not None and plan_async:
            import threading

            def make():
                try:
                    self._plan = um.plan(ctx)
                    self._plan.prepare(flow_dist, velocity, 10800.0)      # selective tables ahead of the first call, if known
                except BaseException as exc:      # re-raised by `plan`
                    self._plan_error = exc
            self._plan_thread = threading.Thread(target=make, name='xh-route-plan')
            self._plan_thread.start()
        elif um is not None:
            self._plan = um.plan(ctx)
            self._plan.prepare(flow_dist, velocity, 10800.0)

    @property
    def plan(self):
        """The device routing plan (None without routing)."""
        if self._plan_thread is not None:
            self._plan_thread.join()
            self._plan_thread = None
            if self._plan_error is not None:
                raise self._plan_error
        return self._plan

    # ---- forcing
    def alloc_forcing(self):
        for k in FORCING:
            if k not in self.forcing:
                self.forcing[k] = self.ctx.empty((self.ncell, self.nmonths))
        return self.forcing

    def set_forcing(self, host, tairprev=None):
        """host: dict of [ncell, nmonths] arrays keyed by FORCING (abcd_tmin optional when use_snow is False).
        A read-only memory map of a .npy (np.load(mmap_mode='r'), what DataLoader keeps) is copied straight out of the
        mapping (no host copy: the runtime pins the page-cache pages); tairprev=None leaves the previous-cell temperature to the PM kernel (it reads the row
        above of ``tas``, data_load.py:127-128)."""
        for k in FORCING:
            if k in host and host[k] is not None:
                src = host[k]
                # XH_UPLOAD_FROM_FILE=1: through xh_upload_file (maps the file range itself: 34 GB/s with its own map /
                # unmap per array); default: xh_memcpy_h2d out of numpy's mapping, which stays alive in the loader (51 GB/s)
                where = file_range_of(src) if os.environ.get('XH_UPLOAD_FROM_FILE', '0') == '1' else None
                direct = where is not None
                arr = src if direct else np.asarray(src, dtype=np.float64)
                if arr.shape != (self.ncell, self.nmonths):
                    raise ValueError('forcing {} has shape {}, expected {}'.format(k, arr.shape,
                                                                                   (self.ncell, self.nmonths)))
                if k not in self.forcing:
                    self.forcing[k] = self.ctx.empty((self.ncell, self.nmonths))
                if direct:
                    self.ctx.upload_file(self.forcing[k], where[0], where[1], src.nbytes)
                else:
                    self.forcing[k].upload(arr)
                if k != 'precip':                       # loader transform: everything but precipitation loses its NaNs
                    self.ctx.nan_to_num(self.forcing[k])
        if tairprev is not None:
            self.d_tairprev = self.ctx.nan_to_num(self.ctx.upload(tairprev))

    # ---- stages (asynchronous; call ctx.sync() or download to wait)
    def run_pm(self):
        f = self.forcing
        pm_mod.run_pmpet_device(self.ctx, self.pm_tables, self.ncell, self.start_year, self.end_year, self.water_idx,
                                self.snow_idx, self.lc_years, f['tas'], f['tmin'], f['rhs'], f['wind'], f['rsds'],
                                f['rlds'], self.d_tairprev, self.d_lct, self.d_elev, self.out['pet'])

    def run_abcd(self):
        f = self.forcing
        self.ctx.abcd(self.ncell, self.nmonths, self.abcd_spinup, self.n_groups, self.basin_index, self.par_index,
                      self.npar_rows, self.d_pars, self.out['pet'], f['precip'],
                      f['abcd_tmin'] if self.use_snow else None, self.out['aet'], self.out['q'], self.out['sav'])

    def run_mrtm(self, runoff=None):
        self.ctx.route_series(self.plan, self.nmonths, self.routing_spinup, self.ndays, 10800.0, self.d_flow_dist,
                              self.d_velocity, self.d_area, self.out['q'] if runoff is None else runoff, self.d_S0,
                              self.out['chs'], self.out['avg'], flags=self.route_flags)

    def run_fused(self, with_routing=True, block_months=0, mode=0):
        """PM -> ABCD (-> MRTM) as one pipelined call (xh_run_fused): the stages overlap on the device.  mode 1 ("fed"):
        the routing kernel starts once the first max(spin-ups) months of runoff exist and the rest of PM and ABCD runs beside it."""
        f = self.forcing
        route = with_routing and self.plan is not None
        self.ctx.run_fused(tables=self.pm_tables, ncell=self.ncell, nmonths=self.nmonths, start_year=self.start_year,
                           lc_years=self.lc_years, water_idx=self.water_idx, snow_idx=self.snow_idx, tas=f['tas'],
                           tmin=f['tmin'], rhs=f['rhs'], wind=f['wind'], rsds=f['rsds'], rlds=f['rlds'],
                           tairprev=self.d_tairprev, lct=self.d_lct, elev=self.d_elev, abcd_spinup=self.abcd_spinup,
                           n_groups=self.n_groups, basin_index=self.basin_index, par_index=self.par_index,
                           npar_rows=self.npar_rows, pars=self.d_pars, precip=f['precip'],
                           abcd_tmin=f['abcd_tmin'] if self.use_snow else None, pet=self.out['pet'], aet=self.out['aet'],
                           q=self.out['q'], sav=self.out['sav'], plan=self.plan if route else None,
                           routing_spinup=self.routing_spinup, ndays=self.ndays, dt=10800.0,
                           flow_dist=self.d_flow_dist, velocity=self.d_velocity, area=self.d_area, S0=self.d_S0,
                           chs=self.out['chs'] if route else None, avg=self.out['avg'] if route else None,
                           route_flags=self.route_flags, block_months=block_months, mode=mode)

    def run(self, stages=('pm', 'abcd', 'mrtm'), fused=None, fed=None, after_runoff=None):
        """Enqueue the stages on the context's stream.  With all three stages the default is the FED order (xh_run_fused
        mode 1): the first max(spin-ups) months of PM and ABCD, then the routing kernel, and the remaining
        months of PM and ABCD beside it on a second stream -- identical results, the 2.8 ms of PM + ABCD mostly hidden
        under the routing.  ``fed=False`` (or XH_FEED=0) runs the stages strictly one after the other; ``fused=True`` (or
        XH_FUSED=1) is round 2's block pipeline of PM and ABCD with the routing behind it (slower on MI355X at the full grid).
        ``after_runoff``: called once PET / AET / Q / Sav have been enqueued and before anything waits for the routing --
        the place for a side gather of the four arrays (dist.OutputGather.run_side)."""
        if fused is None:
            fused = os.environ.get('XH_FUSED') == '1'
        if fed is None:
            fed = os.environ.get('XH_FEED', FEED_DEFAULT) == '1'
        whole_years = self.nmonths % 12 == 0
        if fused and 'pm' in stages and 'abcd' in stages and whole_years:
            self.run_fused(with_routing='mrtm' in stages, block_months=int(os.environ.get('XH_FUSED_BLOCK', '0')))
            if after_runoff:
                after_runoff()
            return
        if fed and all(s in stages for s in ('pm', 'abcd', 'mrtm')) and whole_years and self.plan is not None:
            self.run_fused(with_routing=True, mode=1)
            if after_runoff:      # the routing kernel is in the queue; a side gather waits for the runoff's side stream only
                after_runoff()
            return
        if 'pm' in stages:
            self.run_pm()
        if 'abcd' in stages:
            self.run_abcd()
        if after_runoff:
            after_runoff()
        if 'mrtm' in stages:
            self.run_mrtm()

    def download(self, names=OUTPUTS):
        return {k: self.out[k].download() for k in names}

    def download_pinned(self, names=OUTPUTS):
        """All outputs into page-locked host arrays with asynchronous copies and ONE synchronisation (run_model()'s
        result arrays: numpy views of memory the context keeps until it is closed).  XH_PAGEABLE_OUTPUTS=1 falls back to
        plain numpy arrays filled by synchronous copies."""
        if os.environ.get('XH_PAGEABLE_OUTPUTS') == '1':
            return self.download(names)
        host = {k: self.ctx.pinned((self.ncell, self.nmonths)) for k in names}
        self.ctx.sync()                      # settles a routing fault (re-route) before anything is copied
        for k in names:
            self.ctx.d2h_async(host[k], self.out[k])
        self.ctx.sync()
        return host

    def rows(self, darr, cells):
        """Download selected rows of a [ncell, nmonths] device array."""
        cells = np.ascontiguousarray(cells, dtype=np.int64)
        d_idx = self.ctx.upload(cells, dtype=np.int64)
        d_tmp = self.ctx.empty((len(cells), self.nmonths))
        self.ctx.gather_rows(darr, d_idx, len(cells), self.nmonths, d_tmp)
        host = d_tmp.download()
        d_idx.free()
        d_tmp.free()
        return host


def topology_from_world(world):
    """dsid -> upid -> UM for a synth world / DataLoader-like object (coords, flow_dir, nrow, ncol)."""
    from types import SimpleNamespace
    st = SimpleNamespace(ngridrow=world.nrow, ngridcol=world.ncol)
    ds = mrtm_mod.downstream(world.coords, world.flow_dir, st)
    return mrtm_mod.upstream_genmatrix(mrtm_mod.upstream(world.coords, ds, st))


def pipeline_from_world(ctx, world, nmonths, start_year, abcd_spinup, routing_spinup, um=None, **kw):
    tables = pm_mod.tables_from(world, world.nlcs)
    if um is None:
        um = topology_from_world(world)
    return DevicePipeline(ctx, ncell=world.ncell, nmonths=nmonths, start_year=start_year, basin_ids=world.basin_ids,
                          abcd_pars=world.abcd_pars, pm_tables=tables, lct=world.lct, elev=world.elev,
                          lc_years=world.lc_years, um=um, flow_dist=world.flow_dist, velocity=world.velocity,
                          area=world.area, abcd_spinup=abcd_spinup, routing_spinup=routing_spinup, **kw)
