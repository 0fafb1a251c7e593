"""Components: the harness of the PET -> runoff -> routing path (mirror of xanthos/components.py:29-497).

Same selectors (``pet_module = pm``, ``runoff_module = abcd``, ``routing_module = mrtm``), same methods
(``calculate_pet``, ``calculate_runoff``, ``calculate_routing``, ``simulation``, ``calibrate``) and same result
attributes (``PET, AET, Q, Sav, ChStorage, Avg_ChFlow`` as host float64 ``[ncell, nmonths]``, components.py:95-100).
``import_core`` binds the module globals ``pet_mod / runoff_mod / routing_mod`` to this package's plugins, which call
the HIP kernels through the C-ABI.  ``simulation`` keeps the arrays in HBM between the three stages (one upload of
the forcing, one download of the six outputs); the per-stage ``calculate_*`` methods remain for callers that drive
the stages one by one (the calibration routing callback, components.py:486-497).
"""
import logging
import time

import numpy as np

from . import _hip, launch
from .calibrate import calibrate_abcd as calib_mod
from .data_load import DataLoader
from .ini_reader import ValidationException
from .pipeline import DevicePipeline
from .utils import set_month_arrays

pet_mod = runoff_mod = routing_mod = None

_TOPOLOGIES = {}      # (digest of coords + flow directions, grid shape) -> (dsid, upid, UM with its cached device plan)

# result attribute -> name of the array in the device pipeline
_RESULTS = {'PET': 'pet', 'AET': 'aet', 'Q': 'q', 'Sav': 'sav', 'ChStorage': 'chs', 'Avg_ChFlow': 'avg'}


def _result(attr):
    """Host float64 [ncell, nmonths] result attribute (components.py:95-100).  After a device-resident simulation the
    array is fetched from HBM the first time it is read: a run that writes two of the six variables moves two across PCIe."""
    def get(self):
        if attr not in self._host:
            pipe = self.pipe
            key = _RESULTS[attr]
            if pipe is not None and (key in ('pet', 'aet', 'q', 'sav') or pipe.plan is not None):
                t = time.time()
                self._host[attr] = pipe.out[key].download()
                self.timings['download'] = self.timings.get('download', 0.0) + time.time() - t
            else:
                self._host[attr] = np.zeros((self.s.ncell, self.s.nmonths))
        return self._host[attr]

    def put(self, value):
        self._host[attr] = value
    return property(get, put)


class Components:
    PET, AET, Q, Sav = _result('PET'), _result('AET'), _result('Q'), _result('Sav')
    ChStorage, Avg_ChFlow = _result('ChStorage'), _result('Avg_ChFlow')

    def __init__(self, config):
        self.s = config
        self.import_core()
        self.timings = {}                  # seconds per phase of run_model(): load, topology, plan, upload, kernels, download, ...
        t0 = time.time()
        self.data = DataLoader(config)
        self.timings['load'] = time.time() - t0
        self.yr_imth_dys = set_month_arrays(self.s.nmonths, self.s.StartYear, self.s.EndYear)
        self.routing_timestep_hours = 3 * 3600          # seconds, despite the name (components.py:91)
        self._host = {}                    # result arrays already on the host (the rest: zeros, or still in HBM)
        self.pipe = None                   # DevicePipeline of the last device-resident simulation
        self.um = self.dsid = self.upid = None
        self.instream_flow = None
        self._writer = None                # OutWriter of output_simulation(): q / ac come from it
        self._q = self._ac = None
        # several ranks (one process per GPU, started by launch.spawn / any launcher that sets RANK, WORLD_SIZE, ...): the
        # basins are sharded over them and rank 0 ends up with the gathered outputs, writes the files and runs the
        # post-processors; the other ranks have nothing to write
        self.group = launch.current_group()
        self.is_root = self.group is None or self.group.rank == 0
        self.gather = None

    @property
    def q(self):
        """Runoff as written (aggregated / converted), or Q when 'q' is not an output variable (components.py:461-466)."""
        if self._q is None and self._writer is not None:
            self._q = self._writer.get('q') if 'q' in self._writer.output_names else self.Q
        return self._q

    @property
    def ac(self):
        """Channel flow as written, or Avg_ChFlow (components.py:467-472)."""
        if self._ac is None and self._writer is not None:
            self._ac = self._writer.get('avgchflow') if 'avgchflow' in self._writer.output_names else self.Avg_ChFlow
        return self._ac

    def import_core(self):
        """Bind the selected plugins (components.py:114-142)."""
        global pet_mod, runoff_mod, routing_mod
        if self.s.pet_module == 'pm':
            from .pet import penman_monteith as pet_mod
        elif self.s.pet_module != 'none':
            raise ValidationException("pet_module '{}' is not part of the MI355X hot path".format(self.s.pet_module))
        if self.s.runoff_module == 'abcd':
            from .runoff import abcd as runoff_mod
        elif self.s.runoff_module != 'none':
            raise ValidationException("runoff_module '{}' is not part of the MI355X hot path".format(self.s.runoff_module))
        if self.s.routing_module == 'mrtm':
            from .routing import mrtm as routing_mod

    # ------------------------------------------------------------------ stage by stage (host arrays)
    def calculate_pet(self):
        """Monthly PET (components.py:189-210)."""
        if self.s.pet_module == 'pm':
            return pet_mod.run_pmpet(self.data, self.s.ncell, self.s.pm_nlcs, self.s.StartYear, self.s.EndYear,
                                     self.s.pm_water_idx, self.s.pm_snow_idx, self.s.pm_lc_years, device=self.s.device)
        if self.s.pet_module == 'none':
            return self.data.pet_out

    def calculate_runoff(self, step_num=None, pet=None):
        """ABCD over all months (components.py:212-247)."""
        if self.s.runoff_module == 'abcd':
            rg = runoff_mod.abcd_execute(n_basins=self.s.n_basins, basin_ids=self.data.basin_ids, pet=pet,
                                         precip=self.data.precip, tmin=self.data.tmin, calib_file=self.s.calib_file,
                                         n_months=self.s.nmonths, spinup_steps=self.s.runoff_spinup, jobs=self.s.ro_jobs)
            self.PET, self.AET, self.Q, self.Sav = rg
        elif getattr(self.s, 'alt_runoff', None) is not None:
            self.Q = np.load(self.s.alt_runoff)

    def topology(self):
        """dsid -> upid -> UM, built once per Components (the reference rebuilds it on every call, :268-270) -- and once
        per process for one grid: the UM of the last two distinct (coords, flow directions) is kept with its device
        routing plan, so a second run_model() on the same grid (a scenario sweep) does not partition the networks again."""
        if self.um is None:
            import hashlib
            h = hashlib.blake2b(digest_size=16)
            for a in (self.data.coords, self.data.flow_dir):
                h.update(np.ascontiguousarray(a, dtype=np.float64).tobytes())
            key = (h.hexdigest(), int(self.s.ngridrow), int(self.s.ngridcol))
            hit = _TOPOLOGIES.get(key)
            if hit is None:
                dsid = routing_mod.downstream(self.data.coords, self.data.flow_dir, self.s)
                upid = routing_mod.upstream(self.data.coords, dsid, self.s)
                hit = (dsid, upid, routing_mod.upstream_genmatrix(upid))
                while len(_TOPOLOGIES) >= 2:
                    _TOPOLOGIES.pop(next(iter(_TOPOLOGIES)))
                _TOPOLOGIES[key] = hit
            self.dsid, self.upid, self.um = hit
        return self.um

    def route_flags(self):
        """[[mrtm]] routing_form -> flag of xh_route_series (0: the library default)."""
        return {'reassociated': _hip.XH_ROUTE_REASSOC, 'exact': _hip.XH_ROUTE_EXACT}.get(getattr(self.s, 'routing_form', 'default'), 0)

    def calculate_routing(self, runoff):
        """Spin-up + simulation of MRTM over all months (components.py:249-296). Returns Avg_ChFlow."""
        if self.s.routing_module == 'mrtm':
            um = self.topology()
            chs, avg, fend = routing_mod.route_series(um, self.data.flow_dist, self.data.str_velocity, self.data.area,
                                                      runoff, self.yr_imth_dys[:, 2], self.s.routing_spinup,
                                                      S0=self.data.chs_prev, dt=self.routing_timestep_hours,
                                                      device=self.s.device, flags=self.route_flags())
            self.ChStorage, self.Avg_ChFlow, self.instream_flow = chs, avg, fend
            return self.Avg_ChFlow

    # ------------------------------------------------------------------ whole simulation, device resident
    def simulation(self, run_pet=True, run_runoff=True, run_routing=True, pet_num_steps=0, runoff_num_steps=0,
                   routing_num_steps=0, notify='simulation'):
        """Run the configured stages (components.py:298-384)."""
        if self.s.calibrate:
            self.calibrate()
            return
        logging.info('---{} in progress...'.format(notify))
        t0 = time.time()
        s, d = self.s, self.data
        full = (s.pet_module == 'pm' and s.runoff_module == 'abcd' and run_pet and run_runoff)
        if not full:
            pet_out = self.calculate_pet()
            if run_runoff:
                self.calculate_runoff(pet=pet_out)
            if run_routing and s.routing_module == 'mrtm':
                self.calculate_routing(self.Q)
            return
        ctx = _hip.get_context(s.device)
        t = time.time()
        um = self.topology() if (run_routing and s.routing_module == 'mrtm') else None
        self.timings['topology'] = time.time() - t
        if self.group is not None and self.group.size > 1:
            return self._simulation_sharded(ctx, um, t0, notify)
        t = time.time()
        pipe = DevicePipeline(ctx, ncell=s.ncell, nmonths=s.nmonths, start_year=s.StartYear, basin_ids=d.basin_ids,
                              abcd_pars=np.load(s.calib_file) if not isinstance(s.calib_file, np.ndarray) else s.calib_file,
                              pm_tables=pet_mod.tables_from(d, s.pm_nlcs), lct=d.lct_load, elev=d.elev,
                              lc_years=s.pm_lc_years, um=um,
                              flow_dist=d.flow_dist if um is not None else np.zeros(s.ncell),
                              velocity=d.str_velocity if um is not None else np.zeros(s.ncell), area=d.area,
                              abcd_spinup=s.runoff_spinup, routing_spinup=getattr(s, 'routing_spinup', 0),
                              water_idx=s.pm_water_idx, snow_idx=s.pm_snow_idx, use_snow=d.tmin is not None,
                              chs_prev=getattr(d, 'chs_prev', None), plan_async=True, route_flags=self.route_flags())
        ctx.sync()
        self.timings['plan'] = time.time() - t          # static uploads; the routing partition runs on a host thread meanwhile
        t = time.time()
        pipe.set_forcing({'tas': d.tair_load, 'tmin': d.TMIN_load, 'rhs': d.rhs_load, 'wind': d.wind_load,
                          'rsds': d.rsds_load, 'rlds': d.rlds_load, 'precip': d.precip, 'abcd_tmin': d.tmin},
                         tairprev=d._tairprev if hasattr(d, '_tairprev') else d.tairprev_load)
        ctx.sync()
        self.timings['upload'] = time.time() - t
        t = time.time()
        if um is not None:
            pipe.plan                                   # waits for the partition if it is still being made
        self.timings['plan_wait'] = time.time() - t
        t = time.time()
        ctx.timing_reset()
        pipe.run_pm()
        pipe.run_abcd()
        if um is not None:
            pipe.run_mrtm()
        ctx.sync()
        self.timings['kernels'] = time.time() - t
        logging.info('\tPET + runoff + routing kernels: {:.3f} seconds'.format(time.time() - t))
        self._log_stage_rates(ctx, pipe, um is not None)
        self._host = {}                    # the six results stay in HBM until they are read (or written)
        self.pipe = pipe
        self.timings['download'] = 0.0
        logging.info('---{0} has finished successfully: {1} seconds ---'.format(notify, time.time() - t0))

    def _simulation_sharded(self, ctx, um, t0, notify):
        """The device-resident simulation on this rank's share of the grid (components.py:298-384 for whole basins only; the
        reference's only parallel seam is the basin chunking of abcd.py:369-389, whose results it re-scatters by membership --
        the gather below is that step).  ``dist.make_shards`` deals connected components of (same basin) U (flow edge) onto
        the ranks; every rank maps only ITS rows of the forcing files (memory maps: the rows are read, the rest of the files
        is never touched), runs the unchanged pipeline, and the six outputs travel to rank 0 in one RCCL gather (PET / AET /
        Q / Sav beside the routing, ChStorage / Avg_ChFlow behind it), where they sit in HBM in grid order exactly as a
        one-rank run leaves them.  Every rank computes the same partition without talking."""
        from types import SimpleNamespace
        from . import dist
        s, d, group = self.s, self.data, self.group
        t = time.time()
        shards = dist.make_shards(SimpleNamespace(basin_ids=np.asarray(d.basin_ids)), um, group.size)
        c = shards[group.rank].cells
        sub_um = dist.sub_matrix(um, c) if um is not None else None
        pars = np.load(s.calib_file) if not isinstance(s.calib_file, np.ndarray) else s.calib_file
        chs_prev = getattr(d, 'chs_prev', None)
        pipe = DevicePipeline(ctx, ncell=len(c), nmonths=s.nmonths, start_year=s.StartYear, basin_ids=np.asarray(d.basin_ids)[c],
                              abcd_pars=pars, pm_tables=pet_mod.tables_from(d, s.pm_nlcs), lct=np.asarray(d.lct_load)[c],
                              elev=np.asarray(d.elev)[c], lc_years=s.pm_lc_years, um=sub_um,
                              flow_dist=np.asarray(d.flow_dist)[c] if um is not None else np.zeros(len(c)),
                              velocity=np.asarray(d.str_velocity)[c] if um is not None else np.zeros(len(c)),
                              area=np.asarray(d.area)[c], abcd_spinup=s.runoff_spinup,
                              routing_spinup=getattr(s, 'routing_spinup', 0), water_idx=s.pm_water_idx,
                              snow_idx=s.pm_snow_idx, use_snow=d.tmin is not None,
                              chs_prev=None if chs_prev is None else np.asarray(chs_prev)[c], plan_async=True,
                              route_flags=self.route_flags())
        self.timings['plan'] = time.time() - t
        t = time.time()

        def rows(a):
            return None if a is None else np.ascontiguousarray(a[c], dtype=np.float64)      # (a memory map: only these rows are read)
        # the previous CELL's temperature (data_load.py:128-129) of a shard's rows is not the row above: taken from the grid
        prev = c - 1
        tairprev = np.ascontiguousarray(d.tair_load[np.maximum(prev, 0)], dtype=np.float64)
        tairprev[prev < 0] = 0.0
        pipe.set_forcing({'tas': rows(d.tair_load), 'tmin': rows(d.TMIN_load), 'rhs': rows(d.rhs_load), 'wind': rows(d.wind_load),
                          'rsds': rows(d.rsds_load), 'rlds': rows(d.rlds_load), 'precip': rows(d.precip),
                          'abcd_tmin': rows(d.tmin)}, tairprev=tairprev)
        ctx.sync()
        self.timings['upload'] = time.time() - t
        t = time.time()
        names = ('pet', 'aet', 'q', 'sav') + (('chs', 'avg') if um is not None else ())
        gather = dist.OutputGather(ctx, pipe, shards, group, s.ncell, names=names)
        if um is not None:
            pipe.plan
        self.timings['plan_wait'] = time.time() - t
        t = time.time()
        ctx.timing_reset()
        pipe.run(('pm', 'abcd', 'mrtm') if um is not None else ('pm', 'abcd'), fed=False, after_runoff=gather.run_side)
        gather.run_tail()
        ctx.sync()
        group.barrier()
        self.timings['kernels'] = time.time() - t
        logging.info('\tPET + runoff + routing kernels and the gather ({}), {} of {} cells on this rank: {:.3f} seconds'.format(
            gather.kind, len(c), s.ncell, time.time() - t))
        self._log_stage_rates(ctx, pipe, um is not None)
        self._host = {}
        self.gather = gather
        self.shard_pipe = pipe
        # what the result properties and the writer look at: on the root the gathered arrays, in HBM, in grid order
        self.pipe = SimpleNamespace(out=gather.out, plan=pipe.plan, ncell=s.ncell, nmonths=s.nmonths) if self.is_root else None
        self.timings['download'] = 0.0
        logging.info('---{0} has finished successfully: {1} seconds ---'.format(notify, time.time() - t0))

    def _log_stage_rates(self, ctx, pipe, routed):
        """One log line per stage: kernel time (HIP events on the library's stream) and achieved HBM GB/s against the
        algorithmic bytes of the stage (PM 6 reads + 1 write + land cover, ABCD 3 + 3, MRTM 1 + 2)."""
        cm = pipe.ncell * pipe.nmonths
        stages = [('pm_pet', cm * 56 + pipe.ncell * (pipe.nmonths // 12) * self.s.pm_nlcs * 8),
                  ('abcd_spinup', pipe.ncell * pipe.abcd_spinup * 24), ('abcd_sim', cm * 48)]
        if routed:
            stages.append(('mrtm_route', cm * 24 + pipe.ncell * pipe.routing_spinup * 8))
        for name, nbytes in stages:
            ms, n = ctx.timing(name)
            if n:
                logging.info('\t{:12s} {:8.3f} ms, {:7.1f} GB/s of {} MB algorithmic traffic'.format(
                    name, ms / n, nbytes / (ms / n) / 1e6, nbytes // 1000000))
                self.timings['kernel_' + name] = ms / n / 1e3

    def calibrate(self):
        """Calibrate the ABCD parameters per basin (components.py:486-497)."""
        pet_out = self.calculate_pet()
        multi = self.group is not None and self.group.size > 1
        # (several ranks: the basins are dealt over them; the search needs the same explicit seed on every rank)
        calib_mod.calibrate_all(settings=self.s, data=self.data, pet=pet_out, router_function=self.calculate_routing,
                                group=self.group if multi else None, seed=20240807 if multi else None)

    def drought(self):
        """Drought statistics of runoff or soil moisture (components.py:391-399)."""
        if self.s.CalculateDroughtStats and self.is_root:
            from .drought.drought_stats import DroughtStats
            logging.info('---Start Drought Statistics:')
            t0 = time.time()
            DroughtStats(self.s, self.Q, self.Sav)
            logging.info('---Drought Statistics has finished successfully: %s seconds ------' % (time.time() - t0))

    def accessible_water(self):
        """Accessible water per basin (components.py:401-409)."""
        if self.s.CalculateAccessibleWater and self.is_root:
            from .accessible.accessible import AccessibleWater
            logging.info('---Start Accessible Water:')
            t0 = time.time()
            AccessibleWater(self.s, self.data, self.Q)
            logging.info('---Accessible Water has finished successfully: %s seconds ------' % (time.time() - t0))

    def output_simulation(self):
        """Aggregate / convert on the device and write the selected variables (components.py:441-474)."""
        from .data_writer.out_writer import OutWriter
        if not self.is_root:               # (a sharded run: rank 0 holds the gathered outputs and writes)
            return
        names = {'pet': 'PET', 'aet': 'AET', 'q': 'Q', 'soilmoisture': 'Sav', 'avgchflow': 'Avg_ChFlow'}
        # arrays still in HBM go to the writer as they are (it aggregates / converts / saves from there)
        all_outputs = {k: (self.pipe.out[_RESULTS[a]] if self.pipe is not None and a not in self._host
                           and (a != 'Avg_ChFlow' or self.pipe.plan is not None) else getattr(self, a))
                       for k, a in names.items() if k in self.s.output_vars or k in ('q', 'avgchflow')}
        writer = OutWriter(self.s, self.data.area, all_outputs)
        writer.write()
        self._writer, self._q, self._ac = writer, None, None
        q_written = writer.get('q', host=False) if 'q' in writer.output_names else all_outputs['q']
        # always from the written runoff, or from self.Q when 'q' is not among the output variables (:461-472)
        writer.write_aggregates(self.data, q_written, self.s.AggregateRunoffBasin, self.s.AggregateRunoffCountry,
                                self.s.AggregateRunoffGCAMRegion)
