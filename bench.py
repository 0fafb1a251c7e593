#!/usr/bin/env python
"""Benchmark of the Xanthos monthly PET -> runoff -> routing hot path on MI355X.

    python bench.py --gpus N --steps K --warmup W

One "step" = one full pass of the pm_abcd_mrtm pipeline over one synthetic world: Penman-Monteith PET ->
ABCD runoff (spin-up + simulation) -> MRTM routing (spin-up + simulation at 3-hour sub-steps) for 67,420 land cells
x 600 months (BASELINE.json configs[2]; ``--workload pm_abcd`` runs configs[1]).  Forcing is synthetic
(xanthos_amd/synth.py distributions), generated on the device and resident in HBM before the timed region.

N > 1 (launched with torch.distributed.run, one rank per GPU): every rank routes its own 67,420-cell scenario
(different forcing seed) -- weak scaling, no data-path collective; the timed region is bracketed by a barrier and a
device synchronise on both sides and the slowest rank's time is used.  ``--strong`` instead shards the 235 basins of
ONE world over the ranks (BASELINE.json configs[3]) with a single gather of the six outputs to rank 0.

Rank 0 prints ONE JSON line: metric cell-months/s (whole job), the roofline object for the dominant kernel
(HIP-event durations measured on the library's stream inside this run), per-kernel figures, and -- at N = 1 -- the
CPU baseline: the numpy oracle (a port of the reference's algorithm) timed on this host on a bounded sample.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0          # MI355X HBM3E peak (MI355X_MICROARCH.md: 8.0 TB/s spec, ~6.3 TB/s achievable)
NCELL, NBASINS = 67420, 235


def algorithmic_bytes(ncell, nmonths, nlcs, abcd_spinup, routing_spinup):
    """fp64 bytes each kernel must move per launch."""
    cm = ncell * nmonths
    return {
        'pm_pet': cm * (6 * 8 + 8) + ncell * (nmonths // 12) * nlcs * 8,          # 6 forcings + PET + land cover/yr
        'abcd_spinup': ncell * abcd_spinup * 24 + ncell * 48,                      # pet, precip, tmin of spin-up months
        'abcd_sim': cm * 48,                                                       # 3 reads + 3 writes
        'mrtm_route': cm * 24 + ncell * routing_spinup * 8,                        # Q read (+spin-up re-read), 2 writes
    }


def cpu_baseline(pipe, world, args, log):
    """Time the numpy oracle (port of the reference) on a bounded sample; check the GPU outputs on the same cells."""
    from oracle import abcd as o_abcd, mrtm as o_mrtm, pm as o_pm
    from xanthos_amd import synth
    nm, y0 = pipe.nmonths, pipe.start_year
    res, parity = {}, {}

    # ---- PM: a contiguous block of cells (tairprev = previous cell), first pm_years years
    n_pm, pm_years = args.cpu_pm_cells, min(args.cpu_pm_years, nm // 12)
    cells = np.arange(n_pm)
    f = {k: pipe.rows(pipe.forcing[k], cells)[:, :12 * pm_years] for k in ('tas', 'tmin', 'rhs', 'wind', 'rsds', 'rlds')}
    from types import SimpleNamespace
    sub = SimpleNamespace(**{k: getattr(world, k) for k in ('cL', 'beta', 'rslimit', 'ae', 'be', 'Tminopen',
                                                                  'Tminclose', 'VPDclose', 'VPDopen', 'RBLmin',
                                                                  'RBLmax', 'rc', 'emiss', 'alpha', 'lai', 'laimax',
                                                                  'laimin')})
    sub.elev, sub.lct = world.elev[:n_pm], world.lct[:n_pm]
    d = synth.data_bag(sub, f)
    t = time.perf_counter()
    ref_pet = o_pm.run_pmpet(d, n_pm, world.nlcs, y0, y0 + pm_years - 1, 0, 6, world.lc_years)
    t_pm = time.perf_counter() - t
    res['pm'] = n_pm * 12 * pm_years / t_pm
    got = pipe.rows(pipe.out['pet'], cells)[:, :12 * pm_years]

    def gate(x, ref):
        """Worst |x - ref| as a fraction of the north-star tolerance 1e-6 |ref| + 1e-9 (must stay <= 1)."""
        m = ~np.isnan(ref)
        assert np.array_equal(np.isnan(x), np.isnan(ref))
        return float(np.max(np.abs(x[m] - ref[m]) / (1e-6 * np.abs(ref[m]) + 1e-9)))
    parity['pet'] = float(np.nanmax(np.abs(got - ref_pet) / (np.abs(ref_pet) + 1e-9)))
    parity['tolerance_used'] = {'pet': gate(got, ref_pet)}

    # ---- ABCD: whole basins up to ~cpu_abcd_cells cells, full series, joblib threads like the reference
    order = np.argsort(-np.bincount(world.basin_ids, minlength=world.n_basins + 1))
    chosen, count = [], 0
    for b in order[3:]:
        if b == 0:
            continue
        chosen.append(b)
        count += int((world.basin_ids == b).sum())
        if count >= args.cpu_abcd_cells:
            break
    bcells = np.nonzero(np.isin(world.basin_ids, chosen))[0]
    pet_rows = pipe.rows(pipe.out['pet'], bcells)
    pr_rows = pipe.rows(pipe.forcing['precip'], bcells)
    tn_rows = pipe.rows(pipe.forcing['abcd_tmin'], bcells)
    bid = world.basin_ids[bcells]
    remap = {b: i + 1 for i, b in enumerate(sorted(chosen))}
    bid_local = np.array([remap[b] for b in bid])
    pars_local = world.abcd_pars[np.array(sorted(chosen)) - 1]
    t = time.perf_counter()
    aet, q, sav = o_abcd.abcd_parallel(len(chosen), pars_local, bid_local, pet_rows, pr_rows, tn_rows, nm,
                                       pipe.abcd_spinup, jobs=-1)
    t_abcd = time.perf_counter() - t
    res['abcd'] = len(bcells) * nm / t_abcd
    for name, ref in (('aet', aet), ('q', q), ('sav', sav)):
        got = pipe.rows(pipe.out[name], bcells)
        m = ~np.isnan(ref)
        assert np.array_equal(np.isnan(got), np.isnan(ref)), name
        parity[name] = float(np.max(np.abs(got[m] - ref[m]) / (np.abs(ref[m]) + 1e-9)))
        parity['tolerance_used'][name] = gate(got, ref)

    # ---- MRTM: the whole grid (networks cannot be sampled), a few months, scipy CSR like the reference
    if 'mrtm' in args.stages:
        from xanthos_amd.routing import mrtm
        k = args.cpu_mrtm_months
        q_host = pipe.out['q'].download()[:, :k].copy()
        um = pipe.um.tocsr()
        t = time.perf_counter()
        r_chs, r_avg, _ = o_mrtm.route_series(um, world.flow_dist, world.velocity, world.area, q_host, pipe.ndays[:k], 0)
        t_mrtm = time.perf_counter() - t
        res['mrtm'] = world.ncell * k / t_mrtm
        g_chs, g_avg, _ = mrtm.route_series(pipe.um, world.flow_dist, world.velocity, world.area, q_host,
                                            pipe.ndays[:k], 0)
        parity['routing_bit_exact'] = bool(np.array_equal(g_chs, r_chs, equal_nan=True) and
                                           np.array_equal(g_avg, r_avg, equal_nan=True))
    inv = 1.0 / res['pm'] + 1.0 / res['abcd']        # the ABCD rate already includes its spin-up pass
    if 'mrtm' in res:
        inv += (1.0 + pipe.routing_spinup / nm) / res['mrtm']
    value = 1.0 / inv
    sample = ('numpy oracle: PM {} cells x {} months 1 thread = {:.3g} cm/s; ABCD {} cells ({} basins) x {}+{} months '
              'joblib {} threads = {:.3g} cm/s'.format(n_pm, 12 * pm_years, res['pm'], len(bcells), len(chosen), nm,
                                                      pipe.abcd_spinup, os.cpu_count(), res['abcd']))
    if 'mrtm' in res:
        sample += '; MRTM {} cells x {} months 1 thread = {:.3g} cm/s (x{:.2f} for routing spin-up)'.format(
            world.ncell, args.cpu_mrtm_months, res['mrtm'], 1.0 + pipe.routing_spinup / nm)
    sample += '; value = harmonic composition of the stage rates'
    log('cpu baseline: ' + sample)
    log('parity on the sample: ' + json.dumps(parity))
    return {'value': value, 'unit': 'cell-months/s', 'cores': os.cpu_count(), 'kind': 'port', 'sample': sample,
            'stage_rates': res}, parity


FP64_VALU_PEAK_TFLOPS = 78.6    # MI355X vector fp64 peak (MI355X_MICROARCH.md)
CALIB_FLOP_PER_MCM = 120.0      # flop-equivalents per member-cell-month ("Calibration")


def bench_calib(args, ctx, rank, world_size, dist, torch, backend, log):
    """BASELINE configs[4]: one step = one differential-evolution generation (trial vectors, objective of every member
    of every basin over spin-up + simulation months, selection, convergence test) of all 235 basins, entirely on the
    device.  N > 1: the basins are dealt to the ranks by size (strong scaling) and the results gathered at the end."""
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
        table = gather_results(table, owner, dist)
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
    if rank == 0 and world_size == 1 and not args.no_cpu_baseline:
        from oracle import calib as o_calib, de as o_de
        pop, en = de.state(0)
        lo, hi = np.array([b[0] for b in cfg_bounds()]), np.array([b[1] for b in cfg_bounds()])
        order = np.argsort(-cfg.counts)
        picks = [int(order[i * len(order) // args.cpu_calib_evals]) for i in range(args.cpu_calib_evals)]
        cells, t_cpu, worst = 0, 0.0, 0.0
        for i, b in enumerate(picks):
            host = cfg.host_basin(b)
            xs = o_de.scale_parameters(pop[b, i], lo, hi)
            t1 = time.perf_counter()
            ref = o_calib.objective_kge(xs, 0, host['pet'], host['precip'], host['tmin'], args.months, args.abcd_spinup,
                                        'km3_per_mth', host['area'], cfg.obs[b])
            t_cpu += time.perf_counter() - t1
            cells += int(cfg.counts[b])
            worst = max(worst, abs(cfg.evaluate_one(b, xs[None])[0] - ref) / abs(ref))
        cpu = cells * (args.months + args.abcd_spinup) / t_cpu
        result['cpu_baseline'] = {'value': cpu, 'unit': 'member-cell-months/s', 'cores': 1, 'kind': 'port',
                                  'sample': 'numpy oracle objective_kge (basin_runoff + KGE, as the reference evaluates '
                                            'one member at a time): {} evaluations on basins of {}..{} cells, {:.1f} s'
                                            .format(len(picks), int(cfg.counts[picks].min()),
                                                    int(cfg.counts[picks].max()), t_cpu)}
        result['parity'] = {'objective_max_rel_err_vs_oracle': worst}
        result['speedup_vs_cpu_baseline'] = value / cpu
        log('cpu baseline: ' + result['cpu_baseline']['sample'])
    if rank == 0:
        print(json.dumps(result), flush=True)
    cfg.close()
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


def cfg_bounds():
    from xanthos_amd.calibrate.config5 import BOUNDS
    return BOUNDS


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=3)
    ap.add_argument('--warmup', type=int, default=1)
    ap.add_argument('--workload', default='pm_abcd_mrtm', choices=['pm_abcd_mrtm', 'pm_abcd', 'calib'])
    ap.add_argument('--members', type=int, default=512, help='calib: population per basin')
    ap.add_argument('--cpu-calib-evals', type=int, default=24, help='calib: oracle objective evaluations to time')
    ap.add_argument('--months', type=int, default=600)
    ap.add_argument('--start-year', type=int, default=1961)
    ap.add_argument('--abcd-spinup', type=int, default=120)
    ap.add_argument('--routing-spinup', type=int, default=120)
    ap.add_argument('--strong', action='store_true', help='shard ONE world by basins over the ranks (configs[3])')
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--cpu-pm-cells', type=int, default=4096)
    ap.add_argument('--cpu-pm-years', type=int, default=25)
    ap.add_argument('--cpu-abcd-cells', type=int, default=12000)
    ap.add_argument('--cpu-mrtm-months', type=int, default=24)
    ap.add_argument('--route-flags', type=int, default=0)
    args = ap.parse_args()
    args.stages = ('pm', 'abcd', 'mrtm') if args.workload == 'pm_abcd_mrtm' else ('pm', 'abcd')
    if args.workload == 'calib' and args.months == 600:
        args.months = 480                                       # BASELINE configs[4]: 480 + 120 spin-up months

    rank = int(os.environ.get('RANK', '0'))
    world_size = int(os.environ.get('WORLD_SIZE', '1'))
    local_rank = int(os.environ.get('LOCAL_RANK', '0'))
    if world_size != args.gpus:
        if world_size == 1 and args.gpus > 1:
            raise SystemExit('launch with: python -m torch.distributed.run --nnodes=1 --nproc-per-node {} '
                             '--master-addr 127.0.0.1 bench.py --gpus {} ...'.format(args.gpus, args.gpus))
        raise SystemExit('--gpus {} does not match WORLD_SIZE {}'.format(args.gpus, world_size))

    def log(msg):
        if rank == 0:
            print('[bench] ' + msg, file=sys.stderr, flush=True)

    dist = torch = None
    backend = os.environ.get('XH_BENCH_BACKEND', 'nccl')       # "gloo" + XH_BENCH_ONE_DEVICE=1: dry-run of the N > 1
    if os.environ.get('XH_BENCH_ONE_DEVICE') == '1':           # code path with every rank on GPU 0 (1-GPU test boxes)
        local_rank = 0
    if world_size > 1:
        import torch
        import torch.distributed as dist
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        os.environ.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
        torch.cuda.set_device(local_rank)
        dist.init_process_group(backend=backend, rank=rank, world_size=world_size)

    from xanthos_amd import _hip, synth
    from xanthos_amd.pipeline import FORCING, pipeline_from_world, topology_from_world
    ctx = _hip.get_context(local_rank)
    log('device: ' + ctx.name())

    if args.workload == 'calib':
        return bench_calib(args, ctx, rank, world_size, dist, torch, backend, log)

    t0 = time.perf_counter()
    world = synth.make_world()
    um = topology_from_world(world)
    log('synthetic world: {} cells, {} basins, built in {:.1f} s'.format(world.ncell, world.n_basins,
                                                                          time.perf_counter() - t0))
    shard = None
    if args.strong and world_size > 1:
        from xanthos_amd import dist as xdist
        shards = xdist.make_shards(world, um, world_size)
        shard = shards[rank]
        run_world, run_um = xdist.sub_world(world, um, shard)
    else:
        run_world, run_um = world, um
    pipe = pipeline_from_world(ctx, run_world, args.months, args.start_year, args.abcd_spinup, args.routing_spinup,
                               um=run_um, route_flags=args.route_flags)
    info = pipe.plan.info()
    log('routing plan: ' + json.dumps(info))

    # forcing: generated on the device, resident before the timed region
    forcing = pipe.alloc_forcing()
    d_lat = ctx.upload(run_world.latitude)
    seed = synth.MASTER_SEED + 1 + (0 if args.strong else rank)
    if shard is None:
        ctx.synth_forcing(seed, pipe.ncell, pipe.nmonths, d_lat, forcing, nan_frac=0.0)
    else:                                   # generate the full world, keep this rank's rows (incl. tairprev rows)
        xdist.fill_shard_forcing(ctx, world, shard, pipe, seed)
    ctx.sync()

    def barrier():
        if dist is not None:
            dist.barrier()
            torch.cuda.synchronize()
        ctx.sync()

    for _ in range(args.warmup):
        pipe.run(args.stages)
    ctx.sync()
    ctx.timing_reset()

    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        pipe.run(args.stages)
        if shard is not None:
            gathered = xdist.gather_outputs(ctx, pipe, shard, shards, world, dist, torch)
            torch.cuda.synchronize()
            del gathered
    ctx.sync()
    barrier()
    elapsed = time.perf_counter() - t0
    if dist is not None:
        tt = torch.tensor([elapsed], dtype=torch.float64, device='cuda' if backend == 'nccl' else 'cpu')
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        elapsed = float(tt.item())

    units_per_step = NCELL * args.months * (1 if args.strong else world_size)
    value = units_per_step * args.steps / elapsed
    ms_per_step = 1e3 * elapsed / args.steps

    # per-kernel device time from HIP events on the library's stream (this run, timed region only)
    algo = algorithmic_bytes(pipe.ncell, pipe.nmonths, run_world.nlcs, args.abcd_spinup, args.routing_spinup)
    kernels = {}
    for name in ('pm_pet', 'abcd_spinup', 'abcd_basin_mean', 'abcd_sim', 'mrtm_route'):
        ms, n = ctx.timing(name)
        if n:
            k = {'avg_ms': ms / n, 'launches': n}
            if name in algo:
                k['algorithmic_bytes'] = algo[name]
                k['achieved_GBs'] = algo[name] / (ms / n * 1e-3) / 1e9
                k['frac_of_hbm_peak'] = k['achieved_GBs'] / HBM_PEAK_GBS
            kernels[name] = k
    # HBM traffic per launch from the committed rocprofv3 PMC passes of this same command (profiles/roundN/
    # pmc_traffic.json, made by tools/pmc_to_json.py): counters cannot be read from inside the run itself
    traffic, traffic_src = {}, None
    import glob
    for f in sorted(glob.glob(os.path.join(ROOT, 'profiles', 'round*', 'pmc_traffic.json'))):
        try:
            pm = json.load(open(f))
            traffic = {'pm_pet': pm.get('k_pm_pet', {}).get('hbm_bytes'), 'abcd_spinup': pm.get('k_abcd<true>', {}).get('hbm_bytes'),
                       'abcd_sim': pm.get('k_abcd<false>', {}).get('hbm_bytes'),
                       'mrtm_route': (pm.get('k_mrtm_skew') or pm.get('k_mrtm_flow', {})).get('hbm_bytes')}
            traffic_src = os.path.relpath(f, ROOT)
        except (OSError, ValueError):
            pass
    full_config = (args.months == 600 and args.abcd_spinup == 120 and args.routing_spinup == 120 and not args.strong)
    for k, v in traffic.items():
        if k in kernels and v and full_config:
            kernels[k]['traffic_bytes'] = v
    dominant = max((k for k in kernels if k in algo), key=lambda k: kernels[k]['avg_ms'])
    roofline = {'kernel': dominant, 'bound': 'hbm', 'achieved': kernels[dominant]['achieved_GBs'],
                'peak': HBM_PEAK_GBS, 'unit': 'GB/s', 'frac': kernels[dominant]['frac_of_hbm_peak'],
                'traffic': kernels[dominant].get('traffic_bytes'), 'traffic_source': traffic_src,
                'note': 'mrtm_route is bound by the instruction issue of one wave per sub-step (about 36 instructions for the '
                        'median unit, 8-term rows pace the run), not by HBM; see DESIGN.md 4.3'
                if dominant == 'mrtm_route' else ''}
    if dominant == 'mrtm_route':
        nsub = int(sum(int(d * 86400 / 10800) for d in pipe.ndays)) + \
            int(sum(int(d * 86400 / 10800) for d in pipe.ndays[:args.routing_spinup]))
        roofline['substeps'] = nsub
        roofline['us_per_substep'] = kernels['mrtm_route']['avg_ms'] * 1e3 / nsub
        roofline['cell_substeps_per_s'] = pipe.ncell * nsub / (kernels['mrtm_route']['avg_ms'] * 1e-3)
        roofline['cycles_per_substep_at_2.4GHz'] = roofline['us_per_substep'] * 2400.0

    result = {
        'metric': 'cell-months/sec (pm_abcd_mrtm, 67,420 cells)' if args.workload == 'pm_abcd_mrtm'
        else 'cell-months/sec (pm_abcd, 67,420 cells)',
        'value': value, 'unit': 'cell-months/s', 'n_gpus': world_size, 'steps': args.steps, 'warmup': args.warmup,
        'ms_per_step': ms_per_step, 'higher_is_better': True, 'scaling': 'strong' if args.strong else 'weak',
        'vs_baseline': None, 'dtype': 'f64', 'data': 'synthetic',
        'config': {'workload': '{}: {} cells x {} months, {} basins, nlcs {}, abcd spin-up {}, routing spin-up {}, '
                               'dt 10800 s'.format(args.workload, NCELL, args.months, NBASINS, run_world.nlcs,
                                                   args.abcd_spinup, args.routing_spinup),
                   'parallelism': ('basins sharded over {} GPUs + gather'.format(world_size) if args.strong else
                                   '{} independent scenario(s), one per GPU'.format(world_size))},
        'roofline': roofline, 'kernels': kernels, 'routing_plan': info,
    }
    if rank == 0 and world_size == 1 and not args.no_cpu_baseline:
        base, parity = cpu_baseline(pipe, world, args, log)
        result['cpu_baseline'] = base
        result['parity'] = parity
        result['speedup_vs_cpu_baseline'] = value / base['value']
    if rank == 0:
        print(json.dumps(result), flush=True)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
