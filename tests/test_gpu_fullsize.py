"""GPU tests (-m gpu) at BASELINE.json's full sizes: the 67,420-cell grid through every configuration.

* config 3: 600 + 120 months of routing on the full grid, the dataflow kernels against the ORACLE (scipy CSR, ~3 min of
  host time, computed once per module) -- the bit-exact kernels (XH_ROUTE_EXACT, by flag) bit for bit, and what the library
  ships (the reassociated form on the pipeline's prepared plan: single running sums, folded leaves) within its bar -- then, on
  the shipped form: 20 repetitions, a second context routing concurrently on the same device, background load, the
  forced-fault re-route;
* config 1: ``run_model()`` through a generated ``.ini`` on the full grid against the oracle chain;
* config 4: 480 months, the 235 basins in 8 network-closed shards run one after the other on this GPU and reassembled:
  PET / AET / Q / Sav bit-identical to the unsharded run (what rank 0 holds after the gather on an 8-GPU node), the routed
  outputs within the default form's bar (its sums follow the partition).
"""
from types import SimpleNamespace

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def rel(x, ref):
    assert np.array_equal(np.isnan(x), np.isnan(ref))
    m = ~np.isnan(ref)
    return float(np.max(np.abs(x[m] - ref[m]) / (np.abs(ref[m]) + 1e-9))) if m.any() else 0.0


@pytest.fixture(scope='module')
def full():
    """Full-grid pipeline, 600 months, PM -> ABCD run once; the oracle's routing of the run's own runoff."""
    from oracle import mrtm as o_mrtm
    from xanthos_amd import _hip, synth
    from xanthos_amd.pipeline import pipeline_from_world
    ctx = _hip.get_context(0)
    w = synth.make_world()
    nm = 600
    pipe = pipeline_from_world(ctx, w, nm, 1961, 120, 120)
    ctx.synth_forcing(3, w.ncell, nm, ctx.upload(w.latitude), pipe.alloc_forcing(), nan_frac=0.001)
    pipe.run(('pm', 'abcd'), fused=False)
    q = pipe.out['q'].download()
    # river networks dealt over worker processes: every cell's bits are those of the serial month loops
    # (tests/test_oracle_golden.py::test_route_series_by_network_equals_serial), in ~1/10 of their 3 minutes
    import os
    chs, avg, _, _ = o_mrtm.route_series_by_network(pipe.um.tocsr(), w.flow_dist, w.velocity, w.area, q, pipe.ndays, 120,
                                                    n_procs=max(1, min(os.cpu_count() or 1, 16)))
    # 0.1 % of the cells have NaN precipitation -> NaN runoff, which routing carries downstream like the reference
    # (data_load.py:186 keeps precipitation's NaN; mrtm.py has no special case)
    assert np.isnan(q).any() and np.isnan(avg).any() and np.nanmax(avg) > 0 and np.isfinite(avg).mean() > 0.5
    return SimpleNamespace(ctx=ctx, w=w, pipe=pipe, chs=chs, avg=avg)


EXACT = 256      # XH_ROUTE_EXACT


def _check_bits(full, pipe=None, tag=''):
    """The bit-exact kernels: every routed value equal to the oracle's, bit for bit."""
    got = (pipe or full.pipe).download(('chs', 'avg'))
    assert np.array_equal(got['chs'], full.chs, equal_nan=True), tag
    assert np.array_equal(got['avg'], full.avg, equal_nan=True), tag


def _check(full, pipe=None, tag=''):
    """Whatever the library routed with (the default: the reassociated form on the prepared plan; after a forced fault the
    bit-exact workgroup-per-network kernel): identical NaN masks, every value within 1e-9 |ref| + 1e-3 m3 / 1e-9 m3/s.
    Returns the largest relative error over the values that are not tiny."""
    got = (pipe or full.pipe).download(('chs', 'avg'))
    worst = 0.0
    for k, ref, atol in (('chs', full.chs, 1e-3), ('avg', full.avg, 1e-9)):
        x = got[k]
        assert np.array_equal(np.isnan(x), np.isnan(ref)), (tag, k, 'NaN masks')
        m = ~np.isnan(ref)
        err = np.abs(x[m] - ref[m])
        assert (err <= 1e-9 * np.abs(ref[m]) + atol).all(), (tag, k, float(err.max()))
        big = np.abs(ref[m]) > 1e6 * atol
        worst = max(worst, float((err[big] / np.abs(ref[m][big])).max()))
    return worst


def _on_default_plan(pipe, tag=''):
    """The call was routed by k_mrtm_rsum on the PREPARED plan (folded leaves, single sums), no guard trip."""
    ri = pipe.plan.rsum_info()
    assert pipe.plan.info()['last_tree_kernel'] == 4, (tag, pipe.plan.info())
    assert ri['folded'] > 2500 and ri['pair_cells'] > 0 and ri['fold_disabled'] == 0 and ri['units'] <= 1024, (tag, ri)


def test_config3_full_length_routing_equals_oracle(full):
    """67,420 cells x (120 spin-up + 600) months = 175,312 sub-steps: the time-skewed dataflow kernel, the lock-step
    dataflow kernel and the workgroup-per-network kernel are each bit-identical to the oracle (scipy CSR mat-vec)."""
    for flags, kernel in ((0, 2), (8, 1), (4, 0)):
        full.pipe.route_flags = flags | EXACT
        full.pipe.out['chs'].zero()
        full.pipe.out['avg'].zero()
        full.pipe.run_mrtm()
        _check_bits(full, tag=flags)
        assert full.pipe.plan.info()['last_tree_kernel'] == kernel
    # validated mode: the dataflow result cross-checked on the device, bit for bit, against the kernel
    # that needs no ordering assumption between units (one workgroup per network, barriers only)
    from xanthos_amd import _hip
    n_val = full.pipe.plan.info()['validated']
    full.pipe.route_flags = _hip.XH_ROUTE_VALIDATE | EXACT
    full.pipe.out['chs'].zero()
    full.pipe.run_mrtm()
    _check_bits(full, tag='validated')
    assert full.pipe.plan.info()['validated'] == n_val + 1 and full.pipe.plan.info()['last_tree_kernel'] == 2
    # the partition the time-skewed kernel ran on: nearly every lane used (1,054 units would be all of them), far more
    # streams than the 64-cell cut's ~860, all cells in dataflow units
    info = full.pipe.plan.info()
    assert info['flow_cells'] == 67420 and info['fallback_cells'] == 0
    assert 1054 <= info['flow_units'] <= 1075 and info['flow_edges'] > 1200, info
    full.pipe.route_flags = 0


def test_config3_reassociated_routing_within_1e9_of_oracle(full):
    """What the library ships (round 5: the REASSOCIATED form, k_mrtm_rsum: running sums along chains of lanes, fused update;
    round 6: on the pipeline's PREPARED plan -- one running sum per lane, pair units with halos around the cells that may fire
    next to one that may, folded leaves) over 67,420 cells x (120 + 600) months against the oracle: identical NaN masks and
    every one of the 80.9 M routed values within 1e-9 |ref| (+ 1e-3 m3 / 1e-9 m3/s); the north star's gate is 1e-6.  Stage by
    stage, in the fed order (routing fed while PM and ABCD still run), cross-checked on the device against the barrier-only
    bit-exact kernel (XH_ROUTE_VALIDATE compares within 1e-9 for this form) -- and the same on the plan of PAIRS, which is
    what an unprepared plan routes on and where a guard trip falls back to (XH_ROUTE_NO_PLAIN asks for it)."""
    from xanthos_amd import _hip
    full.pipe.route_flags = 0
    for rep in range(3):
        full.pipe.out['chs'].zero()
        full.pipe.out['avg'].zero()
        full.pipe.run_mrtm()
        worst = _check(full, tag=('staged', rep))
        _on_default_plan(full.pipe, ('staged', rep))
    assert worst < 1e-10, worst                       # (measured: 5e-12)
    full.pipe.out['chs'].zero()
    full.pipe.out['avg'].zero()
    full.pipe.run(fed=True)
    _check(full, tag='fed')
    _on_default_plan(full.pipe, 'fed')
    n_val = full.pipe.plan.info()['validated']
    full.pipe.route_flags = _hip.XH_ROUTE_VALIDATE
    full.pipe.out['chs'].zero()
    full.pipe.run_mrtm()
    _check(full, tag='validated')
    _on_default_plan(full.pipe, 'validated')
    assert full.pipe.plan.info()['validated'] == n_val + 1
    # the plan of pairs
    full.pipe.route_flags = _hip.XH_ROUTE_NO_PLAIN
    full.pipe.out['chs'].zero()
    full.pipe.out['avg'].zero()
    full.pipe.run_mrtm()
    assert _check(full, tag='pairs') < 1e-10
    info, ri = full.pipe.plan.info(), full.pipe.plan.rsum_info()
    assert info['last_tree_kernel'] == 4 and ri['pair_cells'] == -1 and ri['folded'] == 0 and ri['fold_disabled'] == 0, (info, ri)
    assert info['flow_cells'] == 67420 and 1054 <= info['flow_units'] <= 1075 and info['skew_max_lag'] <= 96, info
    full.pipe.route_flags = 0


_PREPARED_CHILD = r"""
import json, os, sys
import numpy as np
root, ref_dir = sys.argv[1], sys.argv[2]
sys.path.insert(0, root)
from xanthos_amd import _hip, synth
from xanthos_amd.pipeline import pipeline_from_world
ctx = _hip.get_context(0)
w = synth.make_world()
nm = 600
pipe = pipeline_from_world(ctx, w, nm, 1961, 120, 120)
ctx.synth_forcing(3, w.ncell, nm, ctx.upload(w.latitude), pipe.alloc_forcing(), nan_frac=0.001)
pipe.run(('pm', 'abcd'), fused=False)
q_ref = np.load(os.path.join(ref_dir, 'q.npy'))
refs = {'chs': (np.load(os.path.join(ref_dir, 'chs.npy')), 1e-3), 'avg': (np.load(os.path.join(ref_dir, 'avg.npy')), 1e-9)}
out = {'same_runoff': bool(np.array_equal(pipe.out['q'].download(), q_ref, equal_nan=True))}
def worst(tag):
    got = pipe.download(('chs', 'avg'))
    wr = 0.0
    for k, (ref, atol) in refs.items():
        x = got[k]
        assert np.array_equal(np.isnan(x), np.isnan(ref)), (tag, k, 'NaN masks')
        m = ~np.isnan(ref)
        err = np.abs(x[m] - ref[m])
        assert (err <= 1e-9 * np.abs(ref[m]) + atol).all(), (tag, k, float(err.max()))
        big = np.abs(ref[m]) > 1e6 * atol
        wr = max(wr, float((err[big] / np.abs(ref[m][big])).max()))
    return wr
pipe.out['chs'].zero(); pipe.out['avg'].zero()
pipe.run_mrtm()
ctx.sync()
out['staged'] = worst('staged')
out['info'] = pipe.plan.rsum_info()
out['kernel'] = int(pipe.plan.info()['last_tree_kernel'])
pipe.out['chs'].zero(); pipe.out['avg'].zero()
pipe.run(fed=True)
ctx.sync()
out['fed'] = worst('fed')
out['info_fed'] = pipe.plan.rsum_info()
print(json.dumps(out))
"""


def test_config3_prepared_plan_with_folded_leaves_within_1e9_of_oracle(full, tmp_path):
    """Round 5's prepared plan -- pairs of sums, leaves that do not fire folded into their downstream cells' lanes: 1,012 units,
    every one alone on its SIMD; XH_RSUM_SINGLE=0 asks for it -- at the full grid and length against the oracle: identical NaN
    masks, every routed value within 1e-9 |ref|, stage by stage and in the fed order, the guard quiet.  In a process of its own
    (the switch is read once per process)."""
    import json
    import os
    import subprocess
    import sys
    np.save(tmp_path / 'q.npy', full.pipe.out['q'].download())
    np.save(tmp_path / 'chs.npy', full.chs)
    np.save(tmp_path / 'avg.npy', full.avg)
    script = tmp_path / 'prepared_child.py'
    script.write_text(_PREPARED_CHILD)
    root = os.path.abspath(os.path.join(os.path.dirname(__file__), '..'))
    env = dict(os.environ, XH_FLOW_CHECK='1', XH_RSUM_SINGLE='0')
    for k in ('XH_ROUTE_REASSOC', 'XH_FLOW_FOLD'):
        env.pop(k, None)
    r = subprocess.run([sys.executable, str(script), root, str(tmp_path)], env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-3000:]
    out = json.loads(r.stdout.strip().splitlines()[-1])
    assert out['same_runoff'] and out['kernel'] == 4, out
    assert out['staged'] < 1e-10 and out['fed'] < 1e-10, out                    # (measured: 3e-12)
    for info in (out['info'], out['info_fed']):
        assert info['folded'] > 2500 and info['fold_disabled'] == 0 and info['units'] <= 1024 and info['pair_cells'] == -1, out


def test_config3_all_cell_pm_abcd_parity(full):
    """EVERY cell of the full grid, not a sample: PET of 67,420 cells x 60 months against oracle.pm (penman_monteith.py:394-477)
    and AET / Q / Sav of 67,420 cells x (600 + 120 spin-up) months against oracle.abcd (abcd.py:357-391) fed with the run's own
    PET, on SURVEY 8(d)'s world with 0.1 % NaN-precipitation cells.  The gate is the north star's tolerance
    1e-6 |ref| + 1e-9 with identical NaN masks; values off by more than 1e-9 relative (a flipped branch of a tiered
    function shows up there long before the gate) are bounded too."""
    from oracle import abcd as o_abcd, pm as o_pm
    from xanthos_amd import synth
    w, pipe = full.w, full.pipe
    nm, k = pipe.nmonths, 60

    def gate(name, x, ref, max_flips):
        assert np.array_equal(np.isnan(x), np.isnan(ref)), name + ': NaN masks differ'
        m = ~np.isnan(ref)
        diff = np.abs(x[m] - ref[m])
        beyond = int((diff > 1e-6 * np.abs(ref[m]) + 1e-9).sum())
        flips = int((diff / (np.abs(ref[m]) + 1e-9) > 1e-9).sum())          # bench.py's branch_flip_candidates
        assert beyond == 0, '{}: {} of {} values beyond 1e-6 |ref| + 1e-9 (worst {:.3e})'.format(
            name, beyond, int(m.sum()), float((diff / (1e-6 * np.abs(ref[m]) + 1e-9)).max()))
        assert flips <= max_flips, '{}: {} branch-flip candidates'.format(name, flips)
        return int(m.sum())

    f = {n: pipe.forcing[n].download()[:, :k].copy() for n in ('tas', 'tmin', 'rhs', 'wind', 'rsds', 'rlds')}
    ref_pet = o_pm.run_pmpet(synth.data_bag(w, f), w.ncell, w.nlcs, 1961, 1961 + k // 12 - 1, 0, 6, w.lc_years)
    got_pet = pipe.out['pet'].download()
    assert gate('pet', got_pet[:, :k], ref_pet, 20) == w.ncell * k
    del f, ref_pet
    pr, tn = pipe.forcing['precip'].download(), pipe.forcing['abcd_tmin'].download()
    assert np.isnan(pr).any(axis=1).sum() >= 30                    # the NaN-precipitation cells are there
    aet, q, sav = o_abcd.abcd_parallel(w.n_basins, w.abcd_pars, w.basin_ids, got_pet, pr, tn, nm, pipe.abcd_spinup, jobs=-1)
    n = 0
    for name, ref in (('aet', aet), ('q', q), ('sav', sav)):
        n += gate(name, pipe.out[name].download(), ref, 200)
    assert n > 3 * 0.99 * w.ncell * nm                              # 40 M values each, less the NaN cells


def test_config3_fused_pipeline_equals_oracle(full):
    """The pipelined call (PM blocks || ABCD march || routing polling for months) on the full grid, five times in a
    row: PET / AET / Q / Sav identical to the stage-by-stage run, routing within the default form's bar of the oracle."""
    keep = {k: full.pipe.out[k].download() for k in ('pet', 'aet', 'q', 'sav')}
    for rep in range(5):
        for k in full.pipe.out:
            full.pipe.out[k].zero()
        full.pipe.run_fused()
        _check(full, tag=rep)
        for k, ref in keep.items():
            assert np.array_equal(full.pipe.out[k].download(), ref, equal_nan=True), (k, rep)
    assert full.pipe.plan.info()['reroutes'] == 0
    _on_default_plan(full.pipe, 'fused')
    # round 4: the FED order (xh_run_fused mode 1) -- the routing kernel launched after the first 128 months of runoff and
    # fed the other 472 while it runs -- eight times in a row: PET / AET / Q / Sav identical, the routed outputs within the bar,
    # no re-route, every call on the prepared plan
    n0 = full.ctx.timing('feed_gate')[1]
    for rep in range(8):
        for k in full.pipe.out:
            full.pipe.out[k].zero()
        full.pipe.run(fed=True, fused=False)
        _check(full, tag=('fed', rep))
        for k, ref in keep.items():
            assert np.array_equal(full.pipe.out[k].download(), ref, equal_nan=True), (k, rep)
    assert full.ctx.timing('feed_gate')[1] == n0 + 8
    assert full.pipe.plan.info()['reroutes'] == 0
    _on_default_plan(full.pipe, 'fed')


def test_config3_twenty_repetitions_and_background_load(full):
    """No state leaks between launches (stream rings and counters are reused 20 times), also while another context
    keeps the GPU busy with PM / ABCD kernels (uneven timing between producers and consumers)."""
    from xanthos_amd import _hip
    from xanthos_amd.pipeline import pipeline_from_world
    other = _hip.Context(0)
    bg = pipeline_from_world(other, full.w, 120, 1961, 30, 0)
    other.synth_forcing(4, full.w.ncell, 120, other.upload(full.w.latitude), bg.alloc_forcing(), nan_frac=0.0)
    for rep in range(20):
        full.pipe.out['chs'].zero()
        full.pipe.out['avg'].zero()
        if rep % 5 == 4:
            for _ in range(40):
                bg.run(('pm', 'abcd'))
        full.pipe.run_mrtm()
        _check(full, tag=rep)
        _on_default_plan(full.pipe, rep)
        other.sync()
    assert full.pipe.plan.info()['reroutes'] == 0
    other.close()


def test_config3_two_contexts_route_concurrently(full):
    """Two contexts (two streams, as two host threads or two processes would own) route the full grid on device 0 at
    the same time, three rounds: the dataflow units of both launches cannot all be resident, so either the launches
    serialise or a bounded wait times out and the library re-routes with the workgroup-per-network kernel -- in both
    cases both contexts end with the oracle's values and no error."""
    from xanthos_amd import _hip
    from xanthos_amd.pipeline import pipeline_from_world
    other = _hip.Context(0)
    p2 = pipeline_from_world(other, full.w, 600, 1961, 120, 120)
    other._check(_hip.lib().xh_memcpy_d2d(other.handle, p2.out['q'].ptr, full.pipe.out['q'].ptr, p2.out['q'].nbytes))
    other.sync()
    for rep in range(3):
        for p in (full.pipe, p2):
            p.out['chs'].zero()
            p.out['avg'].zero()
        full.ctx.sync()
        other.sync()
        full.pipe.run_mrtm()              # asynchronous: both kernels are in flight together
        p2.run_mrtm()
        _check(full, tag=('a', rep))
        _check(full, pipe=p2, tag=('b', rep))
    print('re-routes: ctx A %d, ctx B %d' % (full.pipe.plan.info()['reroutes'], p2.plan.info()['reroutes']))
    other.close()


def test_config3_forced_fault_is_rerouted(full):
    """XH_ROUTE_TEST_FAULT makes unit 0 of the dataflow kernel raise the device fault word as a timed-out wait would:
    the next synchronising call re-runs the routing with one workgroup per network and the caller sees valid outputs
    (XH_OK when nothing else was enqueued behind the routing, XH_ERR_DEVICE with valid routing outputs otherwise).
    After a fault the plan's next 8 calls skip the dataflow kernels (a device that stays shared must not cost a
    timeout per call); then they are tried again, and a fault-free call resets the back-off."""
    from xanthos_amd import _hip
    from xanthos_amd.pipeline import pipeline_from_world
    # a pipeline (= routing plan) of its own: the back-off must not leak into the tests that share `full`
    pipe = pipeline_from_world(full.ctx, full.w, 600, 1961, 120, 120)
    full.ctx._check(_hip.lib().xh_memcpy_d2d(full.ctx.handle, pipe.out['q'].ptr, full.pipe.out['q'].ptr,
                                             pipe.out['q'].nbytes))

    def backed_off_then_back(tag):
        skipped = 0
        while True:
            pipe.out['chs'].zero()
            pipe.run_mrtm()
            _check(full, pipe=pipe, tag=(tag, skipped))
            if pipe.plan.info()['last_tree_kernel'] == 4:      # the dataflow kernel of the default form is back
                return skipped
            skipped += 1
            assert skipped <= 8, 'the dataflow kernels never came back'

    before = pipe.plan.info()['reroutes']
    for flags in (_hip.XH_ROUTE_TEST_FAULT, _hip.XH_ROUTE_TEST_FAULT | _hip.XH_ROUTE_NO_SKEW):
        pipe.out['chs'].zero()
        pipe.out['avg'].zero()
        pipe.route_flags = flags
        pipe.run_mrtm()
        pipe.route_flags = 0
        _check(full, pipe=pipe, tag=flags)                    # the download is the synchronising call
        assert backed_off_then_back(flags) == 8
    assert pipe.plan.info()['reroutes'] == before + 2
    # two faulting calls in flight, then a kernel that read the invalid outputs: the error is reported, not lost
    pipe.out['chs'].zero()
    pipe.route_flags = _hip.XH_ROUTE_TEST_FAULT
    pipe.run_mrtm()
    pipe.route_flags = 0
    pipe.run_mrtm()                                           # gives up at once: the fault word is sticky
    tmp = full.ctx.empty((pipe.ncell, 50))
    full.ctx.agg_time(pipe.ncell, 600, 12, 0, None, pipe.out['avg'], tmp)
    with pytest.raises(_hip.HipError, match='must be recomputed'):
        full.ctx.sync()
    _check(full, pipe=pipe, tag='after error')                # routing outputs were recomputed all the same
    assert pipe.plan.info()['reroutes'] == before + 4
    assert backed_off_then_back('recovered') == 8             # one fault event, however many calls it hit
    # a copy enqueued behind a faulting call is "later work" too (gather / scatter / async copies bump the sequence)
    rows = full.ctx.upload(np.arange(100, dtype=np.int64), dtype=np.int64)      # (a synchronous upload would settle the fault)
    picked = full.ctx.empty((100, 600))
    pipe.route_flags = _hip.XH_ROUTE_TEST_FAULT
    pipe.run_mrtm()
    pipe.route_flags = 0
    full.ctx.gather_rows(pipe.out['avg'], rows, 100, 600, picked)
    with pytest.raises(_hip.HipError, match='must be recomputed'):
        full.ctx.sync()
    # freeing an input of a faulting call settles (re-routes) it first instead of leaving a dangling pointer behind
    pipe.out['chs'].zero()
    pipe.route_flags = _hip.XH_ROUTE_TEST_FAULT
    pipe.run_mrtm()
    pipe.route_flags = 0
    n0 = pipe.plan.info()['reroutes']
    scratch = full.ctx.empty((16,))
    scratch.free()                                            # xh_free -> xh_settle -> re-route
    assert pipe.plan.info()['reroutes'] == n0 + 1
    _check(full, pipe=pipe, tag='after free')
    for a in (tmp, rows, picked):
        a.free()


def test_config1_run_model_full_grid(tmp_path):
    """BASELINE config 1's workload on HIP: run_model() through a generated pm_abcd_mrtm .ini on the full 67,420-cell
    grid (3 years: ABCD needs >= 25 spin-up months, abcd.py:258-266; config 1's 12 months are the first year), every
    output against the oracle chain."""
    from oracle import abcd as o_abcd, months as o_months, mrtm as o_mrtm, pm as o_pm
    from xanthos_amd import run_model, synth
    w = synth.make_world()
    nm = 36
    f = synth.make_forcing(w, nm)
    ini = synth.write_example(str(tmp_path), w, f, 1971, 1973, runoff_spinup=25, routing_spinup=12)
    res = run_model(ini)
    assert res.Q.shape == (67420, nm) and res.Avg_ChFlow.shape == (67420, nm)
    d = synth.data_bag(w, f)
    pet = o_pm.run_pmpet(d, w.ncell, w.nlcs, 1971, 1973, 0, 6, w.lc_years)
    assert rel(res.PET, pet) < 1e-9
    _, aet, q, sav = o_abcd.abcd_execute(w.n_basins, w.basin_ids, res.PET, f['precip'], np.nan_to_num(f['abcd_tmin']),
                                         w.abcd_pars, nm, 25, -1)
    # chained run at full size: the ABCD cancellation (y = rpt - sqrt(...)) leaves up to ~5e-8; the gate is 1e-6
    assert rel(res.AET, aet) < 2e-7 and rel(res.Q, q) < 2e-7 and rel(res.Sav, sav) < 2e-7
    assert np.nanmin(res.Q) >= 0 and np.isnan(res.Q).any()                  # NaN-precipitation cells stay NaN
    st = SimpleNamespace(ngridrow=w.nrow, ngridcol=w.ncol)
    um = o_mrtm.upstream_genmatrix(o_mrtm.upstream(w.coords, o_mrtm.downstream(w.coords, w.flow_dir, st), st))
    ndays = o_months.set_month_arrays(nm, 1971, 1973)[:, 2]
    chs, avg, _ = o_mrtm.route_series(um, res.data.flow_dist, res.data.str_velocity, res.data.area, res.Q, ndays, 12)
    ref = SimpleNamespace(chs=chs, avg=avg)
    got = SimpleNamespace(download=lambda keys: {'chs': res.ChStorage, 'avg': res.Avg_ChFlow})
    assert _check(ref, pipe=got, tag='run_model') < 1e-10
    _on_default_plan(res.pipe, 'run_model')


def test_config4_eight_shards_480_months():
    """BASELINE config 4's workload emulated on one GPU: 1971-2010, the 235 basins packed into 8 network-closed shards,
    each shard's pipeline run on this GPU with shard-local forcing generation, outputs reassembled in grid order:
    PET / AET / Q / Sav bit-identical to the unsharded run, ChStorage / Avg_ChFlow within the default form's bar."""
    from xanthos_amd import _hip, synth
    from xanthos_amd.dist import fill_shard_forcing, make_shards, sub_world
    from xanthos_amd.pipeline import OUTPUTS, pipeline_from_world, topology_from_world
    ctx = _hip.get_context(0)
    w = synth.make_world()
    um = topology_from_world(w)
    nm, seed = 480, 41
    whole = pipeline_from_world(ctx, w, nm, 1971, 120, 120, um=um)
    d_lat = ctx.upload(w.latitude)
    ctx.synth_forcing(seed, w.ncell, nm, d_lat, whole.alloc_forcing(), nan_frac=0.001)
    whole.run()
    ref = whole.download()
    for a in list(whole.out.values()) + list(whole.forcing.values()):
        a.free()
    shards = make_shards(w, um, 8)
    sizes = np.array([len(s.cells) for s in shards])
    assert sizes.min() > 0 and sizes.max() - sizes.min() < 0.05 * w.ncell          # LPT balance
    seen = np.zeros(w.ncell, dtype=bool)
    for s in shards:
        sw, sum_ = sub_world(w, um, s)
        pipe = pipeline_from_world(ctx, sw, nm, 1971, 120, 120, um=sum_)
        fill_shard_forcing(ctx, w, s, pipe, seed, nan_frac=0.001)
        pipe.run()
        out = pipe.download()
        for k in OUTPUTS:
            if k in ('chs', 'avg'):      # (the default form's sums follow the partition: a shard's is not the world's)
                x, r, atol = out[k], ref[k][s.cells], 1e-3 if k == 'chs' else 1e-9
                assert np.array_equal(np.isnan(x), np.isnan(r)), (k, s.rank)
                m = ~np.isnan(r)
                assert (np.abs(x[m] - r[m]) <= 1e-9 * np.abs(r[m]) + atol).all(), (k, s.rank)
            else:
                assert np.array_equal(out[k], ref[k][s.cells], equal_nan=True), (k, s.rank)
        assert pipe.plan.info()['last_tree_kernel'] == 4 and pipe.plan.rsum_info()['fold_disabled'] == 0, s.rank
        seen[s.cells] = True
        for a in list(pipe.out.values()) + list(pipe.forcing.values()) + [pipe.d_tairprev]:
            a.free()
        pipe.plan.close()
    assert seen.all()
