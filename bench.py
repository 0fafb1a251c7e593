#!/usr/bin/env python
"""Benchmark of the Xanthos monthly PET -> runoff -> routing hot path on MI355X.

    python bench.py --gpus N --steps K --warmup W

One "step" = one full pass of the pm_abcd_mrtm pipeline over one synthetic world: Penman-Monteith PET ->
ABCD runoff (spin-up + simulation) -> MRTM routing (spin-up + simulation at 3-hour sub-steps) for 67,420 land cells
x 600 months (BASELINE.json configs[2]; ``--workload pm_abcd`` runs configs[1]).  Forcing is synthetic
(xanthos_amd/synth.py distributions), generated on the device and resident in HBM before the timed region.

N > 1, one rank process per GPU -- started by torch.distributed.run, or by this script itself when it is run plainly as
``python bench.py --gpus N`` (it then starts the N rank processes BEFORE anything touches a GPU, relays rank 0's JSON
line and exits with the worst exit code): the 235 basins of ONE world are sharded over the ranks (BASELINE.json
configs[3]; strong scaling) and every step ends with a single gather of the six outputs to rank 0 -- PET / AET / Q / Sav
travel on a second stream while the routing runs; the timed region is bracketed by a barrier and a device synchronise
on both sides and the slowest rank's time is used.
The figure for N independent whole-world scenarios (one per GPU, no collective) is measured afterwards and printed as
the secondary object ``replicas``; ``--replicas`` makes that mode the primary one (weak scaling).
``--workload calib`` runs BASELINE configs[4] (one step = one differential-evolution generation of all 235 basins).

Rank 0 prints ONE JSON line: metric cell-months/s (whole job), the roofline object for the dominant kernel
(HIP-event durations measured on the library's stream inside this run), per-kernel figures, and -- at N = 1 -- the
CPU baseline: the numpy oracle (a port of the reference's algorithm) timed on this host.  The run is also a GATE: it exits
with code 3 (after printing the line, which says why in ``gate``) when a value lies beyond the north-star tolerance,
a NaN pattern differs or the routed series is not bit-identical to the oracle's.
"""
import argparse
import json
import os
import subprocess
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0          # MI355X HBM3E peak (MI355X_MICROARCH.md: 8.0 TB/s spec, ~6.3 TB/s achievable)
SHADER_CLOCK_HZ = 2.4e9        # MI355X peak engine clock (MI355X_MICROARCH.md); s_memtime showed 2.37 GHz under this load
# lone-wave cost of one routing sub-step, nothing else on the device: a (2,3) row in the bit-exact pair form
# (tools/micro/substep_plain.hip, round 3), the reassociated form with both of its reads (tools/micro/substep_rsum.hip,
# round 5; profiles/round5/substep_rsum.txt), its single-sum form (8-byte entries, 8 fp64 operations: round 6, same tool, mode 6;
# profiles/round6/substep_rsum.txt)
SUBSTEP_FLOOR_CYCLES = {'pair': 186.0, 'reassoc': 104.0, 'single': 77.0}
LDS_BYTES_PER_CLK = 128.0      # per CU (MI355X_MICROARCH.md, LDS)
ROUTE_KERNELS = {4: 'k_mrtm_rsum', 2: 'k_mrtm_wave', 3: 'k_mrtm_skew', 1: 'k_mrtm_flow'}
NCELL, NBASINS = 67420, 235


class TorchGroup:
    """``torch.distributed`` behind the small process-group interface the package's multi-GPU code asks for (rank, size,
    bcast, allreduce, gather, barrier: ``xanthos_amd/launch.py``, whose own ``SocketGroup`` is what ``run_model()`` uses).  The
    bench keeps torch.distributed because the driver starts it under ``torch.distributed.run`` and its timing contract is
    written in those terms; the group only ever carries ids, flags and -- in the host fall-back of the gather -- rows."""

    def __init__(self, dist, torch, backend):
        self.dist, self.torch = dist, torch
        self.rank, self.size = dist.get_rank(), dist.get_world_size()
        self.dev = 'cuda' if backend == 'nccl' else 'cpu'

    def bcast(self, obj, src=0):
        box = [obj]
        self.dist.broadcast_object_list(box, src=src)
        return box[0]

    def allreduce(self, x, op='max'):
        t = self.torch.tensor([float(x)], dtype=self.torch.float64, device=self.dev)
        self.dist.all_reduce(t, op={'max': self.dist.ReduceOp.MAX, 'min': self.dist.ReduceOp.MIN, 'sum': self.dist.ReduceOp.SUM}[op])
        v = float(t.item())
        return int(v) if isinstance(x, int) else v

    def gather(self, obj, root=0, raw=False):
        got = [None] * self.size if self.rank == root else None
        self.dist.gather_object(bytes(obj) if raw else obj, got, dst=root)
        return got

    def barrier(self):
        self.dist.barrier()


def algorithmic_bytes(ncell, nmonths, nlcs, abcd_spinup, routing_spinup):
    """fp64 bytes each kernel must move per launch."""
    cm = ncell * nmonths
    return {
        'pm_pet': cm * (6 * 8 + 8) + ncell * (nmonths // 12) * nlcs * 8,          # 6 forcings + PET + land cover/yr
        'abcd_spinup': ncell * abcd_spinup * 24 + ncell * 48,                      # pet, precip, tmin of spin-up months
        'abcd_sim': cm * 48,                                                       # 3 reads + 3 writes
        'mrtm_route': cm * 24 + ncell * routing_spinup * 8,                        # Q read (+spin-up re-read), 2 writes
    }


def cpu_mrtm_child(workdir):
    """``python bench.py --cpu-mrtm-child DIR``: the whole routed series on the CPU, in a process that never touches the GPU
    (nothing of xanthos_amd is imported here).  Reads the run's own runoff and the routing inputs from DIR, routes spin-up +
    every month with the oracle -- river networks dealt over worker processes, every cell's bits those of the serial loops
    (oracle/mrtm.py) -- and leaves ChStorage / Avg_ChFlow and the timing there."""
    import scipy.sparse as sparse
    from oracle import mrtm as o_mrtm
    z = np.load(os.path.join(workdir, 'inputs.npz'))
    um = sparse.csr_matrix((z['data'], z['indices'], z['indptr']), shape=(len(z['area']), len(z['area'])))
    q = np.load(os.path.join(workdir, 'q.npy'), mmap_mode='r')
    n_procs = int(z['n_procs'])
    t = time.perf_counter()
    chs, avg, _, cpu = o_mrtm.route_series_by_network(um, z['flow_dist'], z['velocity'], z['area'], q, z['ndays'],
                                                      int(z['spinup']), n_procs=n_procs)
    wall = time.perf_counter() - t
    np.save(os.path.join(workdir, 'chs.npy'), chs)
    np.save(os.path.join(workdir, 'avg.npy'), avg)
    json.dump({'wall_s': wall, 'cpu_s': cpu, 'n_procs': n_procs}, open(os.path.join(workdir, 'timing.json'), 'w'))


def cpu_baseline(pipe, world, args, log):
    """Time the numpy oracle (a port of the reference's algorithm) on this host and check the GPU outputs against it.

    Sizes: PM on the FULL grid for the first ``--cpu-pm-years`` years (it is linear in years), 1 thread
    like the reference; ABCD at FULL size (67,420 cells x all months + spin-up) on joblib threads like the reference's
    ``jobs = -1``; MRTM at FULL size (spin-up + every month of the full grid) in a child process that never touches the
    GPU, the river networks dealt over ``--cpu-mrtm-procs`` worker processes; its rate is quoted per thread (work / summed
    process time of the workers: the reference routes on one thread) and every value of the run's ChStorage / Avg_ChFlow
    is compared bit for bit.  ``--cpu-mrtm-months K`` (K > 0) routes only the first K months instead (quick runs).

    Returns (baseline object, parity object, gate failures)."""
    from types import SimpleNamespace
    from oracle import abcd as o_abcd, mrtm as o_mrtm, pm as o_pm
    from xanthos_amd import synth
    nm, y0 = pipe.nmonths, pipe.start_year
    res, parity, failures = {}, {'tolerance_used': {}}, []

    def report(name, x, ref):
        """SURVEY 8(d) parity gate: max relative error, its 99.999-th percentile, values off by more than 1e-9 relative
        (a flipped branch of a tiered function shows up there long before the 1e-6 gate) and values beyond the gate
        1e-6 |ref| + 1e-9 (the north star's tolerance; any such value, or a differing NaN pattern, fails the run)."""
        nan_equal = bool(np.array_equal(np.isnan(x), np.isnan(ref)))
        parity.setdefault('nan_pattern_equal', {})[name] = nan_equal
        if not nan_equal:
            failures.append('{}: NaN pattern differs from the oracle'.format(name))
        m = ~(np.isnan(ref) | np.isnan(x))
        diff = np.abs(x[m] - ref[m])
        rel = diff / (np.abs(ref[m]) + 1e-9)
        parity[name] = float(rel.max())
        parity['tolerance_used'][name] = float(np.max(diff / (1e-6 * np.abs(ref[m]) + 1e-9)))
        parity.setdefault('p99_999', {})[name] = float(np.percentile(rel, 99.999))
        parity.setdefault('branch_flip_candidates', {})[name] = int((rel > 1e-9).sum())
        beyond = int((diff > 1e-6 * np.abs(ref[m]) + 1e-9).sum())
        parity.setdefault('beyond_gate', {})[name] = beyond
        if beyond:
            failures.append('{}: {} values beyond 1e-6 |ref| + 1e-9'.format(name, beyond))
        parity.setdefault('values_compared', {})[name] = int(m.sum())
        parity.setdefault('nan_values', {})[name] = int(np.isnan(ref).sum())

    # ---- PM: the whole grid, first pm_years years
    pm_years = min(args.cpu_pm_years, nm // 12)
    k = 12 * pm_years
    f = {name: pipe.forcing[name].download()[:, :k].copy() for name in ('tas', 'tmin', 'rhs', 'wind', 'rsds', 'rlds')}
    d = synth.data_bag(world, f)
    t = time.perf_counter()
    ref_pet = o_pm.run_pmpet(d, world.ncell, world.nlcs, y0, y0 + pm_years - 1, 0, 6, world.lc_years)
    t_pm = time.perf_counter() - t
    res['pm'] = world.ncell * k / t_pm
    got_pet = pipe.out['pet'].download()
    report('pet', got_pet[:, :k], ref_pet)
    del f, d, ref_pet

    # ---- ABCD: full size, from the run's own PET (so that the comparison isolates this stage)
    pr, tn = pipe.forcing['precip'].download(), pipe.forcing['abcd_tmin'].download()
    t = time.perf_counter()
    aet, q, sav = o_abcd.abcd_parallel(world.n_basins, world.abcd_pars, world.basin_ids, got_pet, pr, tn, nm,
                                       pipe.abcd_spinup, jobs=-1)
    t_abcd = time.perf_counter() - t
    res['abcd'] = world.ncell * nm / t_abcd
    for name, ref in (('aet', aet), ('q', q), ('sav', sav)):
        report(name, pipe.out[name].download(), ref)
    del aet, sav, pr, tn, got_pet

    # ---- MRTM: the whole grid, scipy CSR like the reference, from the run's own runoff
    mrtm_full = 'mrtm' in args.stages and args.cpu_mrtm_months <= 0
    if 'mrtm' in args.stages:
        q_run = pipe.out['q'].download()
        um = pipe.um.tocsr()
        if mrtm_full:
            import shutil
            import tempfile
            n_procs = args.cpu_mrtm_procs if args.cpu_mrtm_procs > 0 else max(1, min(os.cpu_count() or 1, 16))
            base = '/dev/shm' if os.path.isdir('/dev/shm') and os.access('/dev/shm', os.W_OK) else None
            work = tempfile.mkdtemp(prefix='xh_cpu_mrtm_', dir=base)
            try:
                np.save(os.path.join(work, 'q.npy'), q_run)
                np.savez(os.path.join(work, 'inputs.npz'), data=um.data, indices=um.indices, indptr=um.indptr,
                         flow_dist=world.flow_dist, velocity=world.velocity, area=world.area, ndays=pipe.ndays,
                         spinup=pipe.routing_spinup, n_procs=n_procs)
                env = dict(os.environ)
                env['HIP_VISIBLE_DEVICES'] = env['ROCR_VISIBLE_DEVICES'] = ''      # belt and braces: it imports no GPU code
                for v in ('OMP_NUM_THREADS', 'OPENBLAS_NUM_THREADS', 'MKL_NUM_THREADS'):
                    env[v] = '1'
                child = subprocess.run([sys.executable, os.path.abspath(__file__), '--cpu-mrtm-child', work], env=env)
                if child.returncode != 0:
                    raise RuntimeError('the CPU routing child exited with code {}'.format(child.returncode))
                tm = json.load(open(os.path.join(work, 'timing.json')))
                r_chs, r_avg = np.load(os.path.join(work, 'chs.npy')), np.load(os.path.join(work, 'avg.npy'))
            finally:
                shutil.rmtree(work, ignore_errors=True)
            t_mrtm, t_wall = tm['cpu_s'], tm['wall_s']
            res['mrtm'] = world.ncell * (nm + pipe.routing_spinup) / t_mrtm
            g_chs, g_avg = pipe.out['chs'].download(), pipe.out['avg'].download()
            parity['routing_months_checked'] = '{} spin-up + {} (the whole run, every cell)'.format(pipe.routing_spinup, nm)
            mrtm_note = ('MRTM whole series ({} + {} months) in a child process, {} worker processes over river networks: '
                         '{:.3g} cm/s per thread ({:.1f} s of process time, {:.1f} s wall)'.format(
                             pipe.routing_spinup, nm, n_procs, res['mrtm'], t_mrtm, t_wall))
        else:
            from xanthos_amd.routing import mrtm
            km = min(args.cpu_mrtm_months, nm)
            q_host = q_run[:, :km].copy()
            t = time.perf_counter()
            r_chs, r_avg, _ = o_mrtm.route_series(um, world.flow_dist, world.velocity, world.area, q_host,
                                                  pipe.ndays[:km], 0)
            t_mrtm = time.perf_counter() - t
            res['mrtm'] = world.ncell * km / t_mrtm
            from xanthos_amd import _hip as _h
            g_chs, g_avg, _ = mrtm.route_series(pipe.um, world.flow_dist, world.velocity, world.area, q_host,
                                                pipe.ndays[:km], 0, flags=_h.XH_ROUTE_EXACT)
            parity['routing_months_checked'] = '{} (no spin-up); the default checks the whole run'.format(km)
            mrtm_note = 'MRTM {} months 1 thread = {:.3g} cm/s ({:.1f} s), x{:.2f} for routing spin-up'.format(
                km, res['mrtm'], t_mrtm, 1.0 + pipe.routing_spinup / nm)
        from xanthos_amd import _hip
        reassoc_run = mrtm_full and int(pipe.plan.info()['last_tree_kernel']) == 4
        parity['routing_form'] = 'reassociated (k_mrtm_rsum)' if reassoc_run else 'bit-exact'
        if reassoc_run:
            # The run's own outputs came from the reassociated form (the library default): equal to the oracle to rounding.
            # The bar is the one the form was accepted on -- identical NaN masks, every value within 1e-9 |ref| (+ 1e-3 m3 /
            # 1e-9 m3/s) -- far inside the north star's 1e-6; a value beyond it fails the run like a wrong bit did before.
            rr = {}
            for name, g, r, atol in (('chs', g_chs, r_chs, 1e-3), ('avg', g_avg, r_avg, 1e-9)):
                nan_equal = bool(np.array_equal(np.isnan(g), np.isnan(r)))
                m = ~(np.isnan(g) | np.isnan(r))
                err, ref = np.abs(g[m] - r[m]), np.abs(r[m])
                big = ref > 1e6 * atol
                beyond = int((err > 1e-9 * ref + atol).sum())
                rr[name] = {'nan_pattern_equal': nan_equal, 'max_rel': float((err[big] / ref[big]).max()) if big.any() else 0.0,
                            'max_abs': float(err.max()), 'beyond_1e-9': beyond,
                            'beyond_gate_1e-6': int((err > 1e-6 * ref + atol).sum()), 'values_compared': int(m.sum())}
                if not nan_equal:
                    failures.append('routing ({}): NaN pattern differs from the oracle'.format(name))
                if beyond:
                    failures.append('routing ({}): {} values beyond 1e-9 |ref| + {:g}'.format(name, beyond, atol))
            parity['routing_reassociated'] = rr
            # ... and the checker itself: the same runoff through the bit-exact kernel, every value against the oracle's bits
            flags0 = pipe.route_flags
            pipe.route_flags = _hip.XH_ROUTE_EXACT
            pipe.run_mrtm()
            g_chs, g_avg = pipe.out['chs'].download(), pipe.out['avg'].download()
            pipe.route_flags = flags0
            parity['routing_bit_exact_kernel'] = ROUTE_KERNELS.get(int(pipe.plan.info()['last_tree_kernel']), 'k_mrtm_units')
        parity['routing_bit_exact'] = bool(np.array_equal(g_chs, r_chs, equal_nan=True) and
                                           np.array_equal(g_avg, r_avg, equal_nan=True))
        parity['routing_values_compared'] = int(g_chs.size + g_avg.size)
        if not parity['routing_bit_exact']:
            bad = int((~((g_chs == r_chs) | (np.isnan(g_chs) & np.isnan(r_chs)))).sum() +
                      (~((g_avg == r_avg) | (np.isnan(g_avg) & np.isnan(r_avg)))).sum())
            failures.append('routing: {} values of ChStorage / Avg_ChFlow of the bit-exact kernel differ from the oracle'.format(bad))
    inv = 1.0 / res['pm'] + 1.0 / res['abcd']        # the ABCD rate already includes its spin-up pass
    if 'mrtm' in res:
        # the full-size rate counts the spin-up months as work done; the sampled one is scaled by them
        inv += (1.0 + pipe.routing_spinup / nm) / res['mrtm']
    value = 1.0 / inv
    sample = ('numpy oracle on the full 67,420-cell grid: PM {} months 1 thread = {:.3g} cm/s ({:.1f} s); ABCD {}+{} '
              'months joblib {} threads = {:.3g} cm/s ({:.1f} s)'.format(k, res['pm'], t_pm, nm, pipe.abcd_spinup,
                                                                       os.cpu_count(), res['abcd'], t_abcd))
    if 'mrtm' in res:
        sample += '; ' + mrtm_note
    sample += '; value = harmonic composition of the stage rates per simulated cell-month'
    log('cpu baseline: ' + sample)
    log('parity: ' + json.dumps(parity))
    stages = {'pm': 'sample: full grid, {} of {} months (linear in months)'.format(k, nm),
              'abcd': 'full size: {} cells x ({} + {} spin-up) months'.format(world.ncell, nm, pipe.abcd_spinup)}
    if 'mrtm' in res:
        stages['mrtm'] = ('full size: {} spin-up + {} months'.format(pipe.routing_spinup, nm) if mrtm_full else
                          'sample: full grid, {} of {} months, scaled by (1 + spin-up / months)'.format(
                              min(args.cpu_mrtm_months, nm), nm))
    return {'value': value, 'unit': 'cell-months/s', 'cores': os.cpu_count(), 'kind': 'port', 'sample': sample,
            'stage_rates': res, 'stages': stages,
            'full_size': bool(mrtm_full or 'mrtm' not in args.stages),      # full size for ABCD / MRTM
            'full_size_stages': {'pm': bool(k == nm), 'abcd': True, 'mrtm': bool(mrtm_full)}}, parity, failures


def routing_forms(ctx, pipe, log, nsub):
    """Secondary measurement, never `value`: the routing kernel ALONE in both of its forms on the run's own runoff -- the
    reassociated form (XH_ROUTE_REASSOC, the default: k_mrtm_rsum) and the bit-exact one (XH_ROUTE_EXACT: k_mrtm_wave, every unit
    in pair form: the checker) -- each with the figures of merit of DESIGN.md section 5:
    `critical_path` (shader cycles per sub-step of the launch against the lone-wave floor of the form) and, from one more
    launch with the per-unit accounting switched on (XH_FLOW_STATS=1: st[3] holds each unit's LDS operations per sub-step and
    the SIMD it ran on), `lds` = LDS bytes a CU moves per sub-step / (128 B/clk x achieved cycles) and the slowest unit."""
    from xanthos_amd import _hip
    flags0 = pipe.route_flags
    out = {}
    try:
        for name, flag, warm in (('reassociated', _hip.XH_ROUTE_REASSOC, 2), ('bit_exact', _hip.XH_ROUTE_EXACT, 2)):
            pipe.route_flags = flag
            for _ in range(warm):
                pipe.run_mrtm()
                ctx.sync()
            ctx.timing_reset()
            for _ in range(3):
                pipe.run_mrtm()
            ctx.sync()
            ms, n = ctx.timing('mrtm_route')
            ms /= max(n, 1)
            info = pipe.plan.info()
            kern = int(info['last_tree_kernel'])
            single = kern == 4 and pipe.plan.rsum_info()['pair_cells'] >= 0      # the prepared plan: one running sum per lane
            floor = SUBSTEP_FLOOR_CYCLES[('single' if single else 'reassoc') if kern == 4 else 'pair']
            achieved = ms * 1e-3 / nsub * SHADER_CLOCK_HZ
            rec = {'mrtm_route_ms': ms, 'device_kernel': ROUTE_KERNELS.get(kern, 'k_mrtm_units'), 'units': int(info['flow_units']),
                   'streams': int(info['flow_edges']), 'pipeline_depth': int(info['flow_depth']), 'max_lane_lag': int(info['skew_max_lag']),
                   'critical_path': {'substeps': nsub, 'floor_cycles': floor, 'achieved_cycles': achieved, 'frac': floor / achieved,
                                     'clock_hz': SHADER_CLOCK_HZ}}
            if kern == 4:      # which reassociated plan: leaves folded into their downstream cells' lanes (the prepared plan)
                ri = pipe.plan.rsum_info()
                rec['units'] = int(ri['units'])
                rec['folded_leaves'] = int(ri['folded'])
                rec['fold_guard_tripped'] = bool(ri['fold_disabled'])
                rec['plan'] = 'single sums' if single else 'pairs of sums'
                rec['cells_in_pair_units'] = int(ri['pair_cells'])
                if single:      # the few pair units pace the launch: their own floor is the pair form's
                    rec['critical_path']['pacing_units'] = {'form': 'pair units (the cells that may fire next to one that may, '
                                                                    'a CU each)', 'floor_cycles': SUBSTEP_FLOOR_CYCLES['reassoc'],
                                                            'frac': SUBSTEP_FLOOR_CYCLES['reassoc'] / achieved}
            # per-unit accounting: one launch with the statistics on (costs the units a few cycles per check; not timed above)
            os.environ['XH_FLOW_STATS'] = '1'
            try:
                pipe.run_mrtm()
                ctx.sync()
                st = pipe.plan.stats()
            finally:
                os.environ.pop('XH_FLOW_STATS', None)
            if st is not None and len(st):
                raw3 = st[:, 3]
                loop = st[:, 0].astype(np.float64) / nsub
                ops = (raw3 & np.uint64(15)).astype(np.int64)                 # LDS operations per sub-step: reads + the own store
                plain = ((raw3 >> np.uint64(6)) & np.uint64(1)).astype(np.int64)       # bit 6: 8-byte entries (plain / single units)
                lds_bytes = ops * 64 * np.where(plain == 1, 8, 16)             # per unit and sub-step (block transfers: +~3 %)
                if kern == 4 and single:
                    pu = plain == 0
                    rec['pair_units'] = {'n': int(pu.sum()), 'cycles_per_substep_outside_waits': [float(x) for x in np.sort(loop[pu])],
                                         'cycles_per_substep_of_wall': [float(x) for x in np.sort(st[pu, 1].astype(np.float64) / nsub)]}
                hw = (raw3 >> np.uint64(8)) & np.uint64(0xffffffff)
                cu_key = ((raw3 >> np.uint64(40)) & np.uint64(15)).astype(np.int64) * 4096 + ((hw >> np.uint64(8)) & np.uint64(0xff)).astype(np.int64)
                per_cu = np.bincount(np.unique(cu_key, return_inverse=True)[1], weights=lds_bytes)
                slow = int(np.argmax(loop))
                clock = np.median(st[:, 1].astype(np.float64) / (st[:, 2].astype(np.float64) / 100e6))
                ach = ms * 1e-3 / nsub * clock                                 # cycles per sub-step at the clock the units saw
                rec['lds'] = {'bytes_per_cu_per_substep_mean': float(per_cu.mean()), 'bytes_per_cu_per_substep_max': float(per_cu.max()),
                              'lds_frac': float(per_cu.mean() / (LDS_BYTES_PER_CLK * ach)),
                              'lds_frac_busiest_cu': float(per_cu.max() / (LDS_BYTES_PER_CLK * ach)),
                              'bytes_per_clk_peak': LDS_BYTES_PER_CLK, 'cycles_per_substep': float(ach),
                              'measured_clock_hz': float(clock)}
                rec['slowest_unit'] = {'cycles_per_substep_outside_waits': float(loop[slow]), 'lds_ops_per_substep': int(ops[slow]),
                                       'reads_per_substep': int(ops[slow]) - 1, 'eight_byte_entries': bool(plain[slow])}
                rec['unit_cycles_per_substep'] = {'median': float(np.median(loop)), 'p90': float(np.percentile(loop, 90)),
                                                  'max': float(loop.max())}
            out[name] = rec
            log('routing alone, {}: {:.2f} ms ({}; {} units), {:.0f} cycles per sub-step against a floor of {:.0f}'.format(
                name, ms, rec['device_kernel'], rec['units'], achieved, floor))
    finally:
        pipe.route_flags = flags0
    out['note'] = ('mrtm_route alone on the device, three calls each after warm-up; `reassociated` is the form the timed steps ran in '
                   'unless routing_plan says otherwise; results of the two forms agree to <= 1e-9 (parity.routing_reassociated), the '
                   'bit-exact one equals the oracle bit for bit (parity.routing_bit_exact)')
    return out


def end_to_end(ctx, pipe, args, log):
    """PCIe-inclusive rate (never `value`): the eight forcing arrays start in page-locked host memory and the six outputs
    end there, as a loader / writer around the boundary would hold them: H2D + loader transform (nan_to_num) + the
    pipeline + D2H, ``steps`` times."""
    from xanthos_amd.pipeline import FORCING, OUTPUTS
    names = [k for k in FORCING if k in pipe.forcing]
    outs = [k for k in OUTPUTS if ('mrtm' in args.stages or k not in ('chs', 'avg'))]
    h_in = {k: ctx.pinned((pipe.ncell, pipe.nmonths)) for k in names}
    h_out = {k: ctx.pinned((pipe.ncell, pipe.nmonths)) for k in outs}
    for k in names:
        pipe.forcing[k].download(h_in[k])
    ctx.sync()
    times = []
    for _ in range(max(args.steps, 2)):
        t0 = time.perf_counter()
        for k in names:
            ctx.h2d_async(pipe.forcing[k], h_in[k])
            if k != 'precip':
                ctx.nan_to_num(pipe.forcing[k])
        pipe.run(args.stages)
        for k in outs:
            ctx.d2h_async(h_out[k], pipe.out[k])
        ctx.sync()
        times.append(time.perf_counter() - t0)
    t = float(np.median(times))
    nbytes = (len(names) + len(outs)) * pipe.ncell * pipe.nmonths * 8
    r = {'value': pipe.ncell * pipe.nmonths / t, 'unit': 'cell-months/s', 'ms_per_step': 1e3 * t,
         'pcie_bytes_per_step': nbytes, 'pcie_GBs_if_serial': nbytes / max(t - args._kernel_s, 1e-9) / 1e9,
         'note': 'pinned-host H2D of {} forcing arrays + nan_to_num + pipeline + D2H of {} outputs, one stream, no '
                 'overlap; median of {} runs'.format(len(names), len(outs), len(times))}
    # A stream of scenarios (ensemble members, one after the other): upload of scenario k + 1, compute of scenario k and
    # download of scenario k - 1 run on three contexts (three streams) of the same device, two buffer sets alternating.
    from xanthos_amd import _hip
    up, dn = _hip.Context(ctx.device), _hip.Context(ctx.device)
    sets = [(pipe.forcing, pipe.out),
            ({k: ctx.empty((pipe.ncell, pipe.nmonths)) for k in names}, {k: ctx.empty((pipe.ncell, pipe.nmonths)) for k in pipe.out})]
    if 'abcd_tmin' not in names and 'abcd_tmin' in pipe.forcing:
        sets[1][0]['abcd_tmin'] = pipe.forcing['abcd_tmin']

    def upload(i):
        for k in names:
            up.h2d_async(sets[i][0][k], h_in[k])
            if k != 'precip':
                up.nan_to_num(sets[i][0][k])

    nscen = max(2 * args.steps, 6)
    upload(0)
    up.sync()
    t0 = time.perf_counter()
    for k in range(nscen + 1):
        cur, prev = k % 2, (k - 1) % 2
        if k + 1 < nscen:
            upload((k + 1) % 2)
        if k < nscen:
            pipe.forcing, pipe.out = sets[cur]
            pipe.run(args.stages)
        if k >= 1:
            for name in outs:
                dn.d2h_async(h_out[name], sets[prev][1][name])
        up.sync()
        ctx.sync()
        dn.sync()
    t_ov = (time.perf_counter() - t0) / nscen
    pipe.forcing, pipe.out = sets[0]
    for a in list(sets[1][0].values()) + list(sets[1][1].values()):
        if a is not pipe.forcing.get('abcd_tmin'):
            a.free()
    up.close()
    dn.close()
    # OutputInYear = 1 (out_writer.py:237-248: annual sums of the outputs): the aggregation runs on the device (xh_agg_time)
    # and only [ncell, years] per output crosses PCIe -- the D2H leg of the serial variant above shrinks 12 x
    nyr = pipe.nmonths // 12
    d_ann = {k: ctx.empty((pipe.ncell, nyr)) for k in outs}
    h_ann = {k: ctx.pinned((pipe.ncell, nyr)) for k in outs}
    times = []
    for _ in range(max(args.steps // 2, 3)):
        t0 = time.perf_counter()
        for k in names:
            ctx.h2d_async(pipe.forcing[k], h_in[k])
            if k != 'precip':
                ctx.nan_to_num(pipe.forcing[k])
        pipe.run(args.stages)
        for k in outs:
            ctx.agg_time(pipe.ncell, pipe.nmonths, 12, 0, None, pipe.out[k], d_ann[k])
            ctx.d2h_async(h_ann[k], d_ann[k])
        ctx.sync()
        times.append(time.perf_counter() - t0)
    t_an = float(np.median(times))
    r['annual_outputs'] = {'value': pipe.ncell * pipe.nmonths / t_an, 'unit': 'cell-months/s', 'ms_per_step': 1e3 * t_an,
                           'pcie_bytes_per_step': (len(names) * pipe.nmonths + len(outs) * nyr) * pipe.ncell * 8,
                           'note': 'as the serial variant, but the six outputs leave as annual sums formed on the device '
                                   '(OutputInYear = 1: xh_agg_time, then {} x [ncell, {}] over PCIe)'.format(len(outs), nyr)}
    for a in d_ann.values():
        a.free()
    for a in h_ann.values():
        ctx.free_pinned(a)
    r['overlapped'] = {'value': pipe.ncell * pipe.nmonths / t_ov, 'unit': 'cell-months/s', 'ms_per_scenario': 1e3 * t_ov,
                       'note': 'a stream of {} scenarios: upload of the next, compute of the current and download of the '
                               'previous one on three streams, two buffer sets'.format(nscen)}
    for a in list(h_in.values()) + list(h_out.values()):
        ctx.free_pinned(a)
    log('end to end (PCIe-inclusive): ' + json.dumps(r))
    return r


FP64_VALU_PEAK_TFLOPS = 78.6    # MI355X vector fp64 peak (MI355X_MICROARCH.md)
CALIB_FLOP_PER_MCM = 120.0      # flop-equivalents per member-cell-month ("Calibration")


def committed_profile(fname):
    """The newest profiles/round*/<fname> (counter passes cannot be taken from inside a run: each figure read from such a
    file carries the file and the date of the pass, and is refused when it is not about the kernel / configuration that ran)."""
    import glob
    for f in reversed(sorted(glob.glob(os.path.join(ROOT, 'profiles', 'round*', fname)))):
        try:
            return json.load(open(f)), os.path.relpath(f, ROOT)
        except (OSError, ValueError):
            continue
    return {}, None


def bench_calib(args, ctx, rank, world_size, dist, torch, backend, log):
    result = calib_measure(args, ctx, rank, world_size, dist, torch, backend, log)
    if rank == 0:
        print(json.dumps(result), flush=True)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


def calib_measure(args, ctx, rank, world_size, dist, torch, backend, log):
    """BASELINE configs[4]: one step = one differential-evolution generation (trial vectors, objective of every member
    of every basin over spin-up + simulation months, selection, convergence test) of all 235 basins, entirely on the
    device.  N > 1: the basins are dealt to the ranks by size (strong scaling) and the results gathered at the end.
    Returns the result line as a dict."""
    from xanthos_amd import synth
    from xanthos_amd.calibrate.calibrate_abcd import assign_basins, gather_results
    from xanthos_amd.calibrate.config5 import Config5
    world = synth.make_world()
    counts = np.bincount(world.basin_ids, minlength=world.n_basins + 1)[1:]
    owner = assign_basins(counts * (args.months + args.abcd_spinup), world_size)
    mine = [b + 1 for b in range(world.n_basins) if owner[b] == rank]
    t0 = time.perf_counter()
    cfg = Config5(ctx, nmembers=args.members, nmonths=args.months, spinup=args.abcd_spinup, seed=synth.MASTER_SEED,
                  world=world, basins=mine)
    log('config 5 on rank 0: {} basins, {} cells, {} members, {}+{} months, set up in {:.1f} s'.format(
        len(mine), int(cfg.counts.sum()), args.members, args.months, args.abcd_spinup, time.perf_counter() - t0))
    de = cfg.de
    de.init()

    def barrier():
        if dist is not None:
            dist.barrier()
            torch.cuda.synchronize()
        ctx.sync()
    # tol = 0: no basin converges, every step is a full generation of every basin
    if args.warmup:
        de.step(args.warmup, tol=0.0)
    ctx.timing_reset()
    barrier()
    t0 = time.perf_counter()
    de.step(args.steps, tol=0.0)
    barrier()
    elapsed = time.perf_counter() - t0
    if dist is not None:
        tt = torch.tensor([elapsed], dtype=torch.float64, device='cuda' if backend == 'nccl' else 'cpu')
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        elapsed = float(tt.item())
    x, fun, nfev, nit, _ = de.result()
    table = np.column_stack([x, fun, nfev, nit])
    if dist is not None:
        table = gather_results(table, owner, TorchGroup(dist, torch, backend))
    total_mcm = args.members * NCELL * (args.months + args.abcd_spinup)
    value = total_mcm * args.steps / elapsed
    ms = {k: ctx.timing(k) for k in ('calib_abcd', 'calib_kge', 'calib_de')}
    kern_ms = ms['calib_abcd'][0] / args.steps            # spin-up march + basin means + simulation march
    local_mcm = cfg.member_cell_months
    achieved = local_mcm * CALIB_FLOP_PER_MCM / (kern_ms * 1e-3) / 1e12
    result = {
        'metric': 'member-cell-months/sec (ABCD DE calibration, {} members x {} basins)'.format(args.members, NBASINS),
        'value': value, 'unit': 'member-cell-months/s', 'n_gpus': world_size, 'steps': args.steps,
        'warmup': args.warmup, 'ms_per_step': 1e3 * elapsed / args.steps, 'higher_is_better': True,
        'scaling': 'strong', 'vs_baseline': None, 'dtype': 'f64', 'data': 'synthetic',
        'config': {'workload': 'calib: differential evolution (best1bin, deferred) of ABCD a,b,c,d,m; {} members x {} '
                               'basins ({} cells) x {}+{} months, set_calibrate 0, km3_per_mth; step = one generation'
                               .format(args.members, NBASINS, NCELL, args.months, args.abcd_spinup),
                   'parallelism': 'basins dealt by size over {} GPU(s), results gathered once'.format(world_size)},
        'objective_evaluations_per_s': args.members * NBASINS * args.steps / elapsed,
        'roofline': {'kernel': 'calib_abcd (k_calib_march spin-up + simulation)', 'bound': 'fp64 valu',
                     'achieved': achieved, 'peak': FP64_VALU_PEAK_TFLOPS, 'unit': 'TFLOP/s',
                     'frac': achieved / FP64_VALU_PEAK_TFLOPS, 'traffic': None,
                     'note': '{} flop-equivalents per member-cell-month (SURVEY 8(d)); no dense contraction, so the '
                             'vector fp64 peak is the ceiling, not MFMA'.format(CALIB_FLOP_PER_MCM)},
        'kernels': {k: {'avg_ms': v[0] / args.steps, 'launches': v[1]} for k, v in ms.items()},
        'host_share_of_step': 1.0 - sum(v[0] for v in ms.values()) / (1e3 * elapsed),
    }
    # MEASURED issue fraction of the two marches: fp64 wave-instructions x 4 cycles over the cycles
    # the chip's SIMDs have in the marches' time.  SQ_INSTS_VALU of k_calib_march_m<true|false> from the committed counter pass
    # of `bench.py --workload calib` (tools/profile_round.sh); the time is this run's.  Refused when the pass is not about
    # this configuration.
    pmc_i, insts_src = committed_profile('pmc_insts.json')
    marches = [pmc_i.get('k_calib_march_m<true>'), pmc_i.get('k_calib_march_m<false>')]
    std_cfg = args.members == 512 and args.months == 480 and args.abcd_spinup == 120 and world_size == 1
    if all(m and m.get('SQ_INSTS_VALU') for m in marches) and std_cfg:
        insts = sum(m['SQ_INSTS_VALU'] for m in marches)
        simds = 4 * int(ctx.cu_count())
        result['roofline']['valu_issue'] = {
            'valu_wave_insts_per_generation': insts, 'cycles_per_wave_inst': 4, 'simds': simds, 'clock_hz': SHADER_CLOCK_HZ,
            'march_ms': kern_ms, 'frac': insts * 4.0 / (simds * kern_ms * 1e-3 * SHADER_CLOCK_HZ),
            # the same against the cycles the marches really had in the counter pass (SQ_BUSY_CYCLES is summed over the chip's 32
            # shader engines: the clock under this fp64 load is ~2.2-2.3 GHz, not the 2.4 GHz of `frac`)
            'frac_at_measured_clock': (insts * 4.0 / (simds * sum(m['SQ_BUSY_CYCLES'] for m in marches) / 32.0)
                                       if all(m.get('SQ_BUSY_CYCLES') for m in marches) else None),
            'wave_insts_per_member_cell_month': insts * 64.0 / local_mcm,
            'source': {'file': insts_src, 'device_kernels': ['k_calib_march_m<true>', 'k_calib_march_m<false>'],
                       'collected': pmc_i.get('_meta', {}).get('collected')},
            'note': 'measured: SQ_INSTS_VALU of both marches x 4 cycles / (SIMDs x their time x clock); calib_abcd also holds the '
                    'basin-mean kernel (microseconds); quarter-rate instructions (one v_rsq_f64 and the conversions of exp() per '
                    'member-cell-month) are counted as 4 cycles like the rest, so the marches are closer to the issue limit than frac says'}
    else:
        result['roofline']['valu_issue'] = {'frac': None, 'refused': 'no counter pass of k_calib_march_m for this configuration in {}'
                                            .format(insts_src)}
    if rank == 0 and world_size == 1 and not args.no_cpu_baseline:
        from oracle import calib as o_calib, de as o_de
        pop, en = de.state(0)
        lo, hi = np.array([b[0] for b in cfg_bounds()]), np.array([b[1] for b in cfg_bounds()])
        order = np.argsort(-cfg.counts)
        picks, cells, t_cpu, worst, worst_gen = [], 0, 0.0, 0.0, 0.0
        for i in range(100000):
            if t_cpu > args.cpu_calib_seconds:
                break
            b = int(order[(i * 37) % len(order)])          # strides through big and small basins alike
            picks.append(b)
            host = cfg.host_basin(b)
            xs = o_de.scale_parameters(pop[b, i % args.members], lo, hi)
            t1 = time.perf_counter()
            ref = o_calib.objective_kge(xs, 0, host['pet'], host['precip'], host['tmin'], args.months, args.abcd_spinup,
                                        'km3_per_mth', host['area'], cfg.obs[b])
            t_cpu += time.perf_counter() - t1
            cells += int(cfg.counts[b])
            if i < 24:
                worst = max(worst, abs(cfg.evaluate_one(b, xs[None])[0] - ref) / abs(ref))
                # ... and the energy the TIMED generations hold for this member (k_calib_march_m: its exp() argument and its
                # groundwater quotient are products with reciprocals -- held to 1e-9, not to the last bits)
                worst_gen = max(worst_gen, abs(en[b, i % args.members] - ref) / abs(ref))
        cpu = cells * (args.months + args.abcd_spinup) / t_cpu
        result['cpu_baseline'] = {'value': cpu, 'unit': 'member-cell-months/s', 'cores': 1, 'kind': 'port',
                                  'sample': 'numpy oracle objective_kge (basin_runoff + KGE, as the reference evaluates '
                                            'one member at a time): {} evaluations on basins of {}..{} cells, {:.1f} s'
                                            .format(len(picks), int(cfg.counts[picks].min()),
                                                    int(cfg.counts[picks].max()), t_cpu)}
        result['parity'] = {'objective_max_rel_err_vs_oracle': worst, 'generation_energy_max_rel_err_vs_oracle': worst_gen,
                            'gate': 1e-9, 'passed': bool(worst <= 1e-9 and worst_gen <= 1e-9)}
        result['speedup_vs_cpu_baseline'] = value / cpu
        log('cpu baseline: ' + result['cpu_baseline']['sample'])
    cfg.close()
    return result


def secondary_lines(ctx, pipe, args, log, kernel_times, pmc_i, insts_src, parity):
    """BASELINE configs[1] and configs[4] inside the default run, so that the driver times them too:
    `pm_abcd` = 10 steps of Penman-Monteith + ABCD on the run's own pipeline and forcing (the parity of their outputs is the
    run's own gate: the same arrays); `calib` = 3 generations of the 512-member x 235-basin calibration (+ 1 warm-up), with the
    objective of a few members checked against the oracle.  Never `value`."""
    import argparse as _ap
    out = {}
    steps = 10
    for _ in range(2):
        pipe.run(('pm', 'abcd'), fed=False)
    ctx.sync()
    ctx.timing_reset()
    t0 = time.perf_counter()
    for _ in range(steps):
        pipe.run(('pm', 'abcd'), fed=False)
    ctx.sync()
    el = time.perf_counter() - t0
    k = kernel_times(steps)
    simds = 4 * int(ctx.cu_count())
    rec = pmc_i.get('k_pm_pet', {})
    pm = k.get('pm_pet', {})
    if rec.get('SQ_INSTS_VALU') and pm:
        pm['valu_frac'] = rec['SQ_INSTS_VALU'] * 4.0 / (simds * pm['avg_ms'] * 1e-3 * SHADER_CLOCK_HZ)
        pm['valu_source'] = {'file': insts_src, 'device_kernel': 'k_pm_pet', 'collected': pmc_i.get('_meta', {}).get('collected')}
    out['pm_abcd'] = {
        'metric': 'cell-months/sec (pm_abcd, 67,420 cells)', 'value': NCELL * args.months * steps / el, 'unit': 'cell-months/s',
        'steps': steps, 'warmup': 2, 'ms_per_step': 1e3 * el / steps, 'dtype': 'f64', 'data': 'synthetic',
        'config': {'workload': 'pm_abcd: {} cells x {} months, {} basins, abcd spin-up {} (BASELINE configs[1])'.format(
            NCELL, args.months, NBASINS, args.abcd_spinup)},
        'roofline': {'pm_pet': {'bound': 'fp64 valu issue', 'valu_frac': pm.get('valu_frac'), 'hbm_frac': pm.get('frac_of_hbm_peak'),
                                'avg_ms': pm.get('avg_ms'), 'source': pm.get('valu_source')},
                     'abcd_sim': {'bound': 'hbm / dependent fp64 chain', 'hbm_frac': k.get('abcd_sim', {}).get('frac_of_hbm_peak'),
                                  'achieved_GBs': k.get('abcd_sim', {}).get('achieved_GBs'), 'avg_ms': k.get('abcd_sim', {}).get('avg_ms')}},
        'kernels': k,
        'parity': {kk: parity[kk] for kk in ('pet', 'aet', 'q', 'sav') if parity and kk in parity} or
                  'PET / AET / Q / Sav of this pipeline are what the run\'s gate holds to the oracle'}
    log('secondary pm_abcd: {:.3f} ms per step'.format(out['pm_abcd']['ms_per_step']))
    # config 5 (frees nothing of the main pipeline: 4.5 GB of 288)
    cargs = _ap.Namespace(**vars(args))
    cargs.workload, cargs.months, cargs.steps, cargs.warmup = 'calib', 480, 3, 1
    cargs.cpu_calib_seconds = 3.0
    try:
        line = calib_measure(cargs, ctx, 0, 1, None, None, None, log)
        out['calib'] = {kk: line[kk] for kk in ('metric', 'value', 'unit', 'steps', 'warmup', 'ms_per_step', 'dtype', 'data', 'config',
                                               'objective_evaluations_per_s', 'roofline', 'kernels', 'parity', 'cpu_baseline')
                        if kk in line}
        log('secondary calib: {:.2f} ms per generation'.format(line['ms_per_step']))
    except Exception as exc:      # never costs the run its line
        out['calib'] = {'error': str(exc)[:300]}
    return out


def cfg_bounds():
    from xanthos_amd.calibrate.config5 import BOUNDS
    return BOUNDS


def launch_ranks(args):
    """``python bench.py --gpus N`` without a launcher: start the N rank processes here, BEFORE anything in this process
    has touched a GPU (never exec, never after a HIP call), one per GPU with RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* set as
    torch.distributed.run would; relay rank 0's JSON line as this process's stdout and return the worst exit code."""
    import socket
    with socket.socket() as sock:
        sock.bind(('127.0.0.1', 0))
        port = sock.getsockname()[1]
    procs = []
    for r in range(args.gpus):
        env = dict(os.environ)
        env.update({'RANK': str(r), 'LOCAL_RANK': str(r), 'WORLD_SIZE': str(args.gpus), 'LOCAL_WORLD_SIZE': str(args.gpus),
                    'MASTER_ADDR': '127.0.0.1', 'MASTER_PORT': str(port)})
        env.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env,
                                      stdout=subprocess.PIPE if r == 0 else subprocess.DEVNULL))
    # a rank that dies leaves the others inside a collective: once one has failed, the rest get a grace period
    import threading
    worst, deadline, got = 0, None, []
    reader = threading.Thread(target=lambda: got.append(procs[0].stdout.read()))      # drained while the ranks run
    reader.start()
    while any(p.poll() is None for p in procs):
        codes = [p.poll() for p in procs]
        if deadline is None and any(c not in (None, 0) for c in codes):
            deadline = time.time() + 30.0
        if deadline is not None and time.time() > deadline:
            for p in procs:
                if p.poll() is None:
                    p.kill()
        time.sleep(0.2)
    reader.join()
    for p in procs:
        worst = max(worst, abs(p.returncode) if p.returncode else 0)
    # stdout carries the ONE JSON line; anything else a rank's libraries printed there (gloo's connection banner) goes to stderr
    for line in (got[0] if got else b'').decode(errors='replace').splitlines():
        (sys.stdout if line.startswith('{') else sys.stderr).write(line + '\n')
    sys.stdout.flush()
    return worst


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=20)
    ap.add_argument('--warmup', type=int, default=5)
    ap.add_argument('--workload', default='pm_abcd_mrtm', choices=['pm_abcd_mrtm', 'pm_abcd', 'calib'])
    ap.add_argument('--members', type=int, default=512, help='calib: population per basin')
    ap.add_argument('--cpu-calib-seconds', type=float, default=15.0, help='calib: seconds of oracle evaluations to time')
    ap.add_argument('--months', type=int, default=600)
    ap.add_argument('--start-year', type=int, default=1961)
    ap.add_argument('--abcd-spinup', type=int, default=120)
    ap.add_argument('--routing-spinup', type=int, default=120)
    ap.add_argument('--replicas', action='store_true',
                    help='N > 1: every rank runs its own whole world (weak scaling) instead of sharding ONE world')
    ap.add_argument('--strong', action='store_true', help='(default for N > 1; kept for compatibility)')
    ap.add_argument('--no-replica-figure', action='store_true', help='N > 1: skip the secondary replica measurement')
    ap.add_argument('--check-gather', action='store_true',
                    help='N > 1: rank 0 also runs the WHOLE world unsharded and compares the gathered arrays bit for bit')
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--no-end-to-end', action='store_true')
    ap.add_argument('--no-secondary', action='store_true',
                    help='skip the secondary lines of the default run (BASELINE configs[1]: 10 steps of pm_abcd; configs[4]: 3 '
                         'generations of the calibration)')
    ap.add_argument('--cpu-full', action='store_true', help='(the default now; kept for compatibility)')
    ap.add_argument('--cpu-pm-years', type=int, default=5)
    ap.add_argument('--cpu-mrtm-months', type=int, default=0,
                    help='0 (default): the CPU baseline routes the WHOLE series in a child process and every routed value '
                         'of the run is checked bit for bit; K > 0: only the first K months (quick runs)')
    ap.add_argument('--cpu-mrtm-procs', type=int, default=0,
                    help='worker processes of the CPU routing (river networks dealt over them); 0 = min(cores, 16)')
    ap.add_argument('--cpu-mrtm-child', default=None, help=argparse.SUPPRESS)
    ap.add_argument('--no-gate', action='store_true', help='report parity failures in the line but exit 0')
    ap.add_argument('--launch-echo', type=int, default=None, help=argparse.SUPPRESS)
    ap.add_argument('--route-flags', type=int, default=0)
    ap.add_argument('--order', default='auto', choices=['auto', 'fed', 'staged'],
                    help='pm_abcd_mrtm: "staged" = the three stages strictly one after the other; "fed" = the routing kernel '
                         'starts after the first max(spin-ups) months and the rest of PM and ABCD runs beside it (xh_run_fused '
                         'mode 1, DESIGN.md 4.7); "auto" = the library default')
    args = ap.parse_args()
    if args.cpu_mrtm_child:                                     # CPU-only child of the baseline leg: no GPU code is imported
        return cpu_mrtm_child(args.cpu_mrtm_child)
    if args.gpus > 1 and 'RANK' not in os.environ and 'WORLD_SIZE' not in os.environ:
        sys.exit(launch_ranks(args))                            # before any GPU call in this process
    if args.launch_echo is not None:                            # (CPU test of the launcher: what a rank was started with)
        if int(os.environ.get('RANK', '0')) == 0:
            print(json.dumps({k: os.environ.get(k) for k in ('RANK', 'LOCAL_RANK', 'WORLD_SIZE', 'MASTER_ADDR', 'MASTER_PORT')}),
                  flush=True)
        sys.exit(args.launch_echo if os.environ.get('RANK') == str(args.gpus - 1) else 0)
    args.stages = ('pm', 'abcd', 'mrtm') if args.workload == 'pm_abcd_mrtm' else ('pm', 'abcd')
    if args.workload == 'calib' and args.months == 600:
        args.months = 480                                       # BASELINE configs[4]: 480 + 120 spin-up months

    rank = int(os.environ.get('RANK', '0'))
    world_size = int(os.environ.get('WORLD_SIZE', '1'))
    local_rank = int(os.environ.get('LOCAL_RANK', '0'))
    if world_size != args.gpus:
        raise SystemExit('--gpus {} does not match WORLD_SIZE {}'.format(args.gpus, world_size))

    def log(msg):
        if rank == 0:
            print('[bench] ' + msg, file=sys.stderr, flush=True)

    dist = torch = None
    backend = os.environ.get('XH_BENCH_BACKEND', 'nccl')       # "gloo" + XH_BENCH_ONE_DEVICE=1: dry-run of the N > 1
    if os.environ.get('XH_BENCH_ONE_DEVICE') == '1':           # code path with every rank on GPU 0 (1-GPU test boxes)
        local_rank = 0
    if world_size > 1:                                         # ranks were started by the launcher before any GPU call
        import torch
        import torch.distributed as dist
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        os.environ.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
        torch.cuda.set_device(local_rank)
        dist.init_process_group(backend=backend, rank=rank, world_size=world_size)

    from xanthos_amd import _hip, synth
    from xanthos_amd.pipeline import pipeline_from_world, topology_from_world
    ctx = _hip.get_context(local_rank)
    log('device: ' + ctx.name())

    if args.workload == 'calib':
        return bench_calib(args, ctx, rank, world_size, dist, torch, backend, log)

    t0 = time.perf_counter()
    world = synth.make_world()
    um = topology_from_world(world)
    log('synthetic world: {} cells, {} basins, built in {:.1f} s'.format(world.ncell, world.n_basins,
                                                                          time.perf_counter() - t0))
    sharded = world_size > 1 and not args.replicas

    def barrier():
        if dist is not None:
            dist.barrier()
            torch.cuda.synchronize()
        ctx.sync()

    def max_over_ranks(x):
        if dist is None:
            return x
        tt = torch.tensor([x], dtype=torch.float64, device='cuda' if backend == 'nccl' else 'cpu')
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        return float(tt.item())

    def whole_world_pipeline(seed):
        pipe = pipeline_from_world(ctx, world, args.months, args.start_year, args.abcd_spinup, args.routing_spinup,
                                   um=um, route_flags=args.route_flags)
        d_lat = ctx.upload(world.latitude)
        ctx.synth_forcing(seed, pipe.ncell, pipe.nmonths, d_lat, pipe.alloc_forcing(), nan_frac=0.001)      # 0.1 % NaN-precipitation cells
        # tairprev[c] = tas[c - 1], zeros for cell 0 (data_load.py:128-129)
        pipe.d_tairprev = ctx.empty((pipe.ncell, pipe.nmonths)).zero()
        ctx._check(_hip.lib().xh_memcpy_d2d(ctx.handle, pipe.d_tairprev.ptr + pipe.nmonths * 8, pipe.forcing['tas'].ptr,
                                            (pipe.ncell - 1) * pipe.nmonths * 8))
        ctx.sync()
        d_lat.free()
        return pipe

    fed = {'auto': None, 'fed': True, 'staged': False}[args.order]

    def timed(pipe, gather=None):
        """Warm-up + the timed region.  Safety net: a device fault in the fed order (a bounded wait of the hand-over timing
        out: never seen since the months-ready word is polled with an atomic, but it would otherwise cost the whole run its
        line) is reported in `stage_order.fed_fault` and the measurement is repeated stage by stage."""
        try:
            return timed_once(pipe, gather)
        except _hip.HipError as exc:
            if order_used['fed'] is False or world_size > 1:
                raise
            log('device fault in the fed order ({}); measuring again stage by stage'.format(str(exc)[:120]))
            order_used['fed'], order_used['fault'] = False, str(exc)[:200]
            try:
                ctx.sync()
            except _hip.HipError:
                pass
            return timed_once(pipe, gather)

    order_used = {'fed': fed, 'fault': None}

    def timed_once(pipe, gather=None):
        # sharded: PET / AET / Q / Sav leave on the gather stream as soon as they are final (beside the routing), ChStorage /
        # Avg_ChFlow behind the routing; every step ends with the gathered arrays complete on rank 0
        fed = order_used['fed']
        side = gather.run_side if gather else None
        for w in range(args.warmup):
            if w < 2 and 'first_calls' in timed.__dict__:      # what a caller who runs ONE step sees (run_model()): see first_calls
                ctx.sync()
                ctx.timing_reset()
                t1 = time.perf_counter()
            pipe.run(args.stages, fed=fed, after_runoff=side)
            if gather:
                gather.run_tail()
            if w < 2 and 'first_calls' in timed.__dict__:
                ctx.sync()
                info_w = pipe.plan.info() if pipe.plan is not None else {}
                timed.first_calls.append({'step_ms': 1e3 * (time.perf_counter() - t1),
                                          'mrtm_route_ms': ctx.timing('mrtm_route')[0],
                                          'cross_checked': int(info_w.get('validated', 0)) > 0 and w == 0})
        ctx.sync()
        ctx.timing_reset()
        barrier()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            pipe.run(args.stages, fed=fed, after_runoff=side)
            if gather:
                gather.run_tail()
        ctx.sync()
        barrier()
        return max_over_ranks(time.perf_counter() - t0)

    timed.first_calls = []      # the first two steps of the first timed() call, each synchronised and timed on its own
    shard = None
    if sharded:
        # BASELINE configs[3]: the 235 basins of ONE world over the ranks, one gather of the six outputs to rank 0
        from xanthos_amd import dist as xdist
        shards = xdist.make_shards(world, um, world_size)
        shard = shards[rank]
        run_world, run_um = xdist.sub_world(world, um, shard)
        pipe = pipeline_from_world(ctx, run_world, args.months, args.start_year, args.abcd_spinup,
                                   args.routing_spinup, um=run_um, route_flags=args.route_flags)
        xdist.fill_shard_forcing(ctx, world, shard, pipe, synth.MASTER_SEED + 1, nan_frac=0.001)      # the whole world's rows
        gather = xdist.OutputGather(ctx, pipe, shards, TorchGroup(dist, torch, backend), world.ncell,
                                    names=('pet', 'aet', 'q', 'sav') + (('chs', 'avg') if 'mrtm' in args.stages else ()))
        elapsed = timed(pipe, gather)
        units_per_step = NCELL * args.months
        parallelism = '{} basins of one world sharded over {} GPUs ({} cells on rank 0), one gather of {} outputs ' \
                      'per step ({})'.format(NBASINS, world_size, pipe.ncell, len(gather.names), gather.kind)
    else:
        run_world = world
        pipe = whole_world_pipeline(synth.MASTER_SEED + 1 + rank)
        elapsed = timed(pipe)
        units_per_step = NCELL * args.months * world_size
        parallelism = ('{} independent scenarios, one per GPU'.format(world_size) if world_size > 1 else
                       'one world on one GPU')
    info = pipe.plan.info() if pipe.plan is not None else {}
    if pipe.plan is not None:
        info['form'] = 'bit-exact kernels, every unit in pair form'
        if int(info.get('last_tree_kernel', 0)) == 4:
            info['form'] = ('reassociated form (k_mrtm_rsum: row sums as running sums along chains of lanes, two LDS reads per '
                            'sub-step for every unit, fused update; equal to the reference to rounding, see parity.routing_reassociated)')
            ri = pipe.plan.rsum_info()
            info['rsum_plan'] = {'kind': 'single sums' if ri['pair_cells'] >= 0 else 'pairs of sums', 'units': int(ri['units']),
                                 'folded_leaves': int(ri['folded']), 'cells_in_pair_units': int(ri['pair_cells']),
                                 'guard_tripped': bool(ri['fold_disabled'])}
        info['guard_trips'] = int(pipe.plan.rsum_info()['guard_trips'])
    log('routing plan: ' + json.dumps(info))
    value = units_per_step * args.steps / elapsed
    ms_per_step = 1e3 * elapsed / args.steps

    # per-kernel device time from HIP events on the library's stream (this run, timed region only)
    algo = algorithmic_bytes(pipe.ncell, pipe.nmonths, run_world.nlcs, args.abcd_spinup, args.routing_spinup)
    def kernel_times(steps):
        """Device time per STEP of each kernel family (a fed step launches PM and the ABCD march twice: two blocks of months)."""
        out = {}
        for name in ('pm_pet', 'abcd_spinup', 'abcd_basin_mean', 'abcd_sim', 'mrtm_route'):
            ms, n = ctx.timing(name)
            if n:
                k = {'avg_ms': ms / steps, 'launches': n, 'launches_per_step': n / steps}
                if name in algo:
                    k['algorithmic_bytes'] = algo[name]
                    k['achieved_GBs'] = algo[name] / (ms / steps * 1e-3) / 1e9
                    k['frac_of_hbm_peak'] = k['achieved_GBs'] / HBM_PEAK_GBS
                out[name] = k
        return out
    kernels = kernel_times(args.steps)
    n_fed = ctx.timing('feed_gate')[1]
    order = {'order': 'fed' if n_fed else 'staged', 'fed_steps': int(n_fed),
             'note': ('the routing kernel is launched after the first max(spin-ups) months of PM and ABCD and fed the other '
                      'months, produced beside it on a second stream (xh_run_fused mode 1): kernel times below overlap, their '
                      'sum exceeds the step' if n_fed else 'the three stages strictly one after the other')}
    if n_fed:
        # PM and ABCD share the chip with the routing kernel in a fed step: their stand-alone times (the figures their
        # rooflines are about) come from three stage-by-stage steps outside the timed region
        gate_ms = ctx.timing('feed_gate')[0] / n_fed
        fed_kernels = kernels
        ctx.timing_reset()
        for _ in range(3):
            pipe.run(args.stages, fed=False)
        ctx.sync()
        alone = kernel_times(3)
        kernels = {k: dict(alone[k]) for k in alone}
        for k in fed_kernels:
            kernels[k]['avg_ms_in_fed_step'] = fed_kernels[k]['avg_ms']
        kernels['mrtm_route'].update({k: v for k, v in fed_kernels['mrtm_route'].items()})      # the dominant kernel: as timed
        kernels['mrtm_route']['avg_ms_alone'] = alone['mrtm_route']['avg_ms']
        order['side_stream_gate_ms'] = gate_ms
    if order_used['fault']:
        order['fed_fault'] = order_used['fault']
    args._kernel_s = sum(k['avg_ms'] for k in kernels.values()) * 1e-3
    # HBM traffic and instruction counts per launch come from the committed rocprofv3 PMC passes of this same command
    # (profiles/roundN/pmc_traffic.json and pmc_insts.json, made by tools/pmc_to_json.py / tools/pmc_insts_json.py):
    # counters cannot be read from inside the run itself.  Each figure carries the device kernel it was taken from and the
    # date of the pass, and is REFUSED (null, with the reason) when that is not the kernel that ran here.
    ran = {'pm_pet': 'k_pm_pet', 'abcd_spinup': 'k_abcd<true>', 'abcd_sim': 'k_abcd_tile<false',
           'mrtm_route': ROUTE_KERNELS.get(int(info.get('last_tree_kernel', 0)), 'k_mrtm_units')}
    full_config = (args.months == 600 and args.abcd_spinup == 120 and args.routing_spinup == 120 and not sharded)

    pmc_t, traffic_src = committed_profile('pmc_traffic.json')
    pmc_i, insts_src = committed_profile('pmc_insts.json')
    traffic_note = {}
    for k, dev in ran.items():
        if k not in kernels:
            continue
        kernels[k]['device_kernel'] = dev
        rec = pmc_t.get(dev)
        if not full_config:
            traffic_note[k] = 'not the configuration the counters were collected on'
        elif not rec or not rec.get('hbm_bytes'):
            traffic_note[k] = 'no counter pass of {} in {}'.format(dev, traffic_src)
        else:
            kernels[k]['traffic_bytes'] = rec['hbm_bytes']
    dominant = max((k for k in kernels if k in algo), key=lambda k: kernels[k]['avg_ms'])
    roofline = {'kernel': dominant, 'bound': 'hbm', 'achieved': kernels[dominant]['achieved_GBs'],
                'peak': HBM_PEAK_GBS, 'unit': 'GB/s', 'frac': kernels[dominant]['frac_of_hbm_peak'],
                'traffic': kernels[dominant].get('traffic_bytes'),
                'traffic_source': {'file': traffic_src, 'device_kernel': ran[dominant],
                                   'collected': pmc_t.get('_meta', {}).get('collected'),
                                   'refused': traffic_note.get(dominant)},
                'note': 'mrtm_route is bound by the sub-step latency / instruction issue of one wave per unit, not by '
                        'HBM (critical_path below is the figure of merit); see DESIGN.md 4.3' if dominant == 'mrtm_route' else ''}
    # Penman-Monteith is bound by fp64 VALU issue -- wave-instructions x 4 cycles / (SIMDs x cycles of the launch)
    if 'pm_pet' in kernels:
        rec = pmc_i.get('k_pm_pet', {})
        simds = 4 * int(ctx.cu_count())
        k = kernels['pm_pet']
        if rec.get('SQ_INSTS_VALU') and full_config:
            k['valu_wave_insts'] = rec['SQ_INSTS_VALU']
            k['valu_frac'] = rec['SQ_INSTS_VALU'] * 4.0 / (simds * k['avg_ms'] * 1e-3 * SHADER_CLOCK_HZ)
            k['lane_insts_per_cell_month'] = rec['SQ_INSTS_VALU'] * 64.0 / (pipe.ncell * pipe.nmonths)
        roofline['pm_pet'] = {'bound': 'fp64 valu issue', 'valu_frac': k.get('valu_frac'),
                              'valu_wave_insts': k.get('valu_wave_insts'), 'cycles_per_wave_inst': 4, 'simds': simds,
                              'clock_hz': SHADER_CLOCK_HZ, 'avg_ms': k['avg_ms'],
                              'source': {'file': insts_src, 'device_kernel': 'k_pm_pet',
                                         'collected': pmc_i.get('_meta', {}).get('collected')}}
    if dominant == 'mrtm_route':
        nsub = int(sum(int(d * 86400 / 10800) for d in pipe.ndays)) + \
            int(sum(int(d * 86400 / 10800) for d in pipe.ndays[:args.routing_spinup]))
        us = kernels['mrtm_route']['avg_ms'] * 1e3 / nsub
        roofline['substeps'] = nsub
        roofline['us_per_substep'] = us
        roofline['cell_substeps_per_s'] = pipe.ncell * nsub / (kernels['mrtm_route']['avg_ms'] * 1e-3)
        # T >= N_substeps x t_substep.  The floor is what ONE sub-step of a lone wave costs with nothing else on
        # the device (tools/micro/substep_plain.hip, MI355X, round 3: a (2,3)-term unit in pair form / a (2,4)-term unit in
        # plain form); achieved = the launch's time per sub-step in shader cycles.
        achieved = us * 1e-6 * SHADER_CLOCK_HZ
        if ran['mrtm_route'] == 'k_mrtm_rsum':
            single = pipe.plan.rsum_info()['pair_cells'] >= 0
            fl = SUBSTEP_FLOOR_CYCLES['single' if single else 'reassoc']
            roofline['critical_path'] = {'substeps': nsub, 'form': 'reassociated, single sums' if single else 'reassociated, pairs of sums',
                                         'floor_cycles': fl, 'achieved_cycles': achieved,
                                         'frac': fl / achieved, 'clock_hz': SHADER_CLOCK_HZ,
                                         'floor_source': 'tools/micro/substep_rsum.hip (lone wave, both reads, no streams; mode 6 for '
                                                         'single sums)'}
            if single:
                roofline['critical_path']['pacing_units'] = {
                    'form': 'the pair units of the plan (the cells that may fire next to one that may and their halos: a CU each)',
                    'floor_cycles': SUBSTEP_FLOOR_CYCLES['reassoc'], 'frac': SUBSTEP_FLOOR_CYCLES['reassoc'] / achieved}
        else:
            roofline['critical_path'] = {'substeps': nsub, 'form': 'bit-exact', 'floor_cycles': SUBSTEP_FLOOR_CYCLES['pair'],
                                         'achieved_cycles': achieved, 'frac': SUBSTEP_FLOOR_CYCLES['pair'] / achieved,
                                         'clock_hz': SHADER_CLOCK_HZ,
                                         'floor_source': 'tools/micro/substep_plain.hip (lone wave, no neighbours, no streams)'}

    result = {
        'metric': 'cell-months/sec (pm_abcd_mrtm, 67,420 cells)' if args.workload == 'pm_abcd_mrtm'
        else 'cell-months/sec (pm_abcd, 67,420 cells)',
        'value': value, 'unit': 'cell-months/s', 'n_gpus': world_size, 'steps': args.steps, 'warmup': args.warmup,
        'ms_per_step': ms_per_step, 'higher_is_better': True, 'scaling': 'weak' if args.replicas else 'strong',
        'vs_baseline': None, 'dtype': 'f64', 'data': 'synthetic',
        'config': {'workload': '{}: {} cells x {} months, {} basins, nlcs {}, abcd spin-up {}, routing spin-up {}, '
                               'dt 10800 s'.format(args.workload, NCELL, args.months, NBASINS, run_world.nlcs,
                                                   args.abcd_spinup, args.routing_spinup),
                   'parallelism': parallelism},
        'roofline': roofline, 'kernels': kernels, 'routing_plan': info, 'stage_order': order,
        # NOT the steady state `value` is about: the first steps of a fresh plan, as run_model()'s single call meets them --
        # the first dataflow call of a plan on a box without a pass on record is cross-checked against the barrier-only
        # kernel (cross_checked: its mrtm_route_ms then contains that second routing), and the plan routes in pair form until
        # the selective plain form has been learnt (routing_plan.form)
        'first_calls': timed.first_calls[:2],
    }
    gate_failed = False
    if sharded:
        result['gather'] = gather.report()
        result['gather_kind'] = result['gather']['kind']
        result['gather_exposed_ms'] = result['gather']['exposed_ms']
        # What one world sharded over N GPUs CAN deliver (terms 1 + 2 + 4): read `value` against this, not
        # against N x the one-GPU line.  The routing of a shard is as many sequential sub-steps as the routing of the world, so
        # term 2 hardly shrinks with N; the hardware's scaling is in `replicas` (independent scenarios) and the calibration
        # fan-out.  Terms from THIS run, rank 0's shard: kernels timed with HIP events, the tail of the gather from the run.
        k_ = kernels
        front = sum(k_[n_]['avg_ms'] for n_ in ('pm_pet', 'abcd_spinup', 'abcd_basin_mean', 'abcd_sim') if n_ in k_)
        lead_months = max(args.abcd_spinup, args.routing_spinup, 1)
        term1 = front * min(1.0, (lead_months + 8) / float(args.months)) if n_fed else front
        term2 = k_['mrtm_route']['avg_ms'] if 'mrtm_route' in k_ else 0.0
        term4 = result['gather']['exposed_ms']
        result['scaling_model'] = {
            'n_gpus': world_size, 'what': 'ONE world, basins sharded: step = PM + ABCD in front of the routing + routing of the '
                                          'slowest rank\'s networks + the gather of ChStorage / Avg_ChFlow behind it',
            'term1_front_ms': term1, 'term2_route_ms': term2, 'term4_gather_tail_ms': term4,
            'modelled_step_ms': (term1 + term2 + (term4 or 0.0) + 0.1) if term4 is not None else None,
            'measured_step_ms': ms_per_step, 'terms_from': 'rank 0 of this run (HIP events)',
            'sequential_substeps': int(sum(int(d * 86400 / 10800) for d in pipe.ndays) +
                                       sum(int(d * 86400 / 10800) for d in pipe.ndays[:args.routing_spinup])) if 'mrtm' in args.stages else 0,
            'note': 'strong scaling of one world is flat by construction (the sub-steps of the routing are sequential whatever the '
                    'cells per GPU: BASELINE.md sections 4 and 7); the north star\'s >= 6 x at 8 GPUs is met by the modes without a '
                    'sequential dependence across ranks: `replicas` below and the calibration fan-out (--workload calib)'}
        if args.check_gather:
            # what rank 0 holds after the gather against the same world run unsharded on its own GPU (same seed: the random
            # streams are keyed on the global cell index), every output, bit for bit
            ok, diff = True, {}
            if rank == 0:
                got = gather.last()
                wpipe = whole_world_pipeline(synth.MASTER_SEED + 1)
                wpipe.run(args.stages)
                ctx.sync()
                # (the reassociated routing form sums along chains the partition lays out, and a shard's partition is not the
                # whole world's: ChStorage / Avg_ChFlow then agree to rounding -- held to 1e-9 here -- not bit for bit; the
                # bit-exact form, --route-flags 256, does not depend on the partition)
                reassoc = int(wpipe.plan.info()['last_tree_kernel']) == 4 if wpipe.plan is not None else False
                for k in gather.names:
                    ref_k = wpipe.out[k].download()
                    same = bool(np.array_equal(got[k], ref_k, equal_nan=True))
                    if not same and reassoc and k in ('chs', 'avg'):
                        m = ~np.isnan(ref_k)
                        same = bool(np.array_equal(np.isnan(got[k]), np.isnan(ref_k)) and
                                    (np.abs(got[k][m] - ref_k[m]) <= 1e-9 * np.abs(ref_k[m]) + (1e-3 if k == 'chs' else 1e-9)).all())
                        diff[k + '_compared'] = 'within 1e-9 (reassociated routing form)'
                    diff[k] = same
                    ok = ok and same
                for a in list(wpipe.out.values()) + list(wpipe.forcing.values()) + [wpipe.d_tairprev]:
                    a.free()
            result['gather']['equals_unsharded'] = ok
            result['gather']['equals_unsharded_by_output'] = diff
            if rank == 0 and not ok:
                gate_failed = not args.no_gate
        if not args.no_replica_figure:
            # secondary figure: N independent whole-world scenarios, one per GPU (no collective on the data path)
            gather.close()
            for a in list(pipe.out.values()) + list(pipe.forcing.values()):
                a.free()
            rpipe = whole_world_pipeline(synth.MASTER_SEED + 1 + rank)
            r_elapsed = timed(rpipe)
            result['replicas'] = {'value': NCELL * args.months * world_size * args.steps / r_elapsed,
                                  'unit': 'cell-months/s', 'ms_per_step': 1e3 * r_elapsed / args.steps,
                                  'scaling': 'weak', 'note': '{} independent scenarios, one per GPU'.format(world_size)}
    if rank == 0 and world_size == 1:
        if 'mrtm' in args.stages and not args.no_end_to_end:
            nsub_all = int(sum(int(d * 86400 / 10800) for d in pipe.ndays)) + \
                int(sum(int(d * 86400 / 10800) for d in pipe.ndays[:args.routing_spinup]))
            result['routing_forms'] = routing_forms(ctx, pipe, log, nsub_all)
        if not args.no_end_to_end:
            result['end_to_end'] = end_to_end(ctx, pipe, args, log)
            pipe.run(args.stages)                      # outputs of the resident run again, for the parity check below
            ctx.sync()
        if not args.no_cpu_baseline:
            base, parity, failures = cpu_baseline(pipe, world, args, log)
            result['cpu_baseline'] = base
            result['parity'] = parity
            result['speedup_vs_cpu_baseline'] = value / base['value']
            # the run is a gate: a value beyond the north-star tolerance, a differing NaN pattern or a routed value that is
            # not the oracle's bit for bit makes the process exit non-zero (after the line, which says why)
            result['gate'] = {'passed': not failures, 'failures': failures,
                              'checks': 'PET / AET / Q / Sav within 1e-6 |ref| + 1e-9 of the oracle with identical NaN '
                                        'patterns; ChStorage / Avg_ChFlow of the run ({}) {}; the bit-exact kernel on the same '
                                        'runoff bit-identical to the oracle ({})'.format(
                                            parity.get('routing_form', 'no routing'),
                                            'within 1e-9 |ref| of the oracle, identical NaN patterns'
                                            if 'routing_reassociated' in parity else 'bit-identical to the oracle',
                                            parity.get('routing_months_checked', 'no routing'))}
            gate_failed = bool(failures) and not args.no_gate
        if args.workload == 'pm_abcd_mrtm' and not args.no_secondary and full_config:
            result['secondary'] = secondary_lines(ctx, pipe, args, log, kernel_times, pmc_i, insts_src, result.get('parity'))
    if dist is not None:
        # ranks the collective layer really sees (a sum over the process group, not the launcher's word for it)
        tt = torch.tensor([1.0], dtype=torch.float64, device='cuda' if backend == 'nccl' else 'cpu')
        dist.all_reduce(tt, op=dist.ReduceOp.SUM)
        result['n_ranks_seen'] = int(tt.item())
    else:
        result['n_ranks_seen'] = 1
    if rank == 0:
        print(json.dumps(result), flush=True)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()
    if gate_failed:
        log('GATE FAILED: ' + '; '.join(result['gate']['failures']))
        sys.exit(3)


if __name__ == '__main__':
    main()
