"""Stress of the fed order's hand-over (xh_run_fused mode 1): many full-grid calls with a first block so small that the
routing units always reach months the side stream has not delivered yet (the library's 128 months) -- every unit parks in the
months-ready wait and has to be woken by the word.  Counts calls, re-routes (bounded-wait timeouts) and mismatches.
python tools/feed_stress.py [calls]"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from xanthos_amd import _hip, synth                      # noqa: E402
from xanthos_amd.pipeline import pipeline_from_world    # noqa: E402

calls = int(sys.argv[1]) if len(sys.argv) > 1 else 100
ctx = _hip.get_context(0)
w = synth.make_world()
pipe = pipeline_from_world(ctx, w, 600, 1961, 120, 120)
lat = ctx.upload(w.latitude)
# two forcings, alternating: a stale staged line left over from the previous call must not be able to pass for the right one
refs = {}
for seed in (3, 4):
    ctx.synth_forcing(seed, w.ncell, 600, lat, pipe.alloc_forcing(), nan_frac=0.001)
    pipe.run(fed=False)
    ctx.sync()
    refs[seed] = {k: pipe.out[k].download() for k in ('q', 'chs', 'avg')}
bad, faults, times = 0, 0, []
for i in range(calls):
    seed = (3, 4)[i & 1]
    ref = refs[seed]
    ctx.synth_forcing(seed, w.ncell, 600, lat, pipe.alloc_forcing(), nan_frac=0.001)
    ctx.sync()
    for k in ('q', 'chs', 'avg'):
        pipe.out[k].zero()
    t = time.perf_counter()
    try:
        pipe.run(fed=True)
        ctx.sync()
    except _hip.HipError as exc:
        faults += 1
        print('call', i, 'fault:', str(exc)[:100], flush=True)
    times.append(time.perf_counter() - t)
    if i % 10 == 0 or times[-1] > 1.0:
        same = all(np.array_equal(pipe.out[k].download(), ref[k], equal_nan=True) for k in ref)
        bad += 0 if same else 1
        print('call {:4d}  {:.1f} ms  identical {}'.format(i, 1e3 * times[-1], same), flush=True)
t = np.array(times) * 1e3
print('calls {}  faults {}  mismatches {}  reroutes {}  ms per call: median {:.2f} p90 {:.2f} max {:.2f}'.format(
    calls, faults, bad, pipe.plan.info()['reroutes'], np.median(t), np.percentile(t, 90), t.max()))
