"""Diagnostic: one world at full size, the stage-by-stage order, then the FED order (xh_run_fused mode 1) call by call with
the library's timers: did the side stream's kernels run beside the routing kernel, how long did its gate wait, what did
the routing kernel cost.  python tools/feed_probe.py [calls]"""
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from xanthos_amd import _hip, synth                      # noqa: E402
from xanthos_amd.pipeline import pipeline_from_world    # noqa: E402

calls = int(sys.argv[1]) if len(sys.argv) > 1 else 3
ctx = _hip.get_context(0)
w = synth.make_world()
pipe = pipeline_from_world(ctx, w, 600, 1961, 120, 120)
ctx.synth_forcing(3, w.ncell, 600, ctx.upload(w.latitude), pipe.alloc_forcing(), nan_frac=0.001)
names = ('pm_pet', 'abcd_spinup', 'abcd_sim', 'mrtm_route', 'feed_gate')


def timers():
    return {k: [round(x, 3) for x in ctx.timing(k)] for k in names}


pipe.run(fed=False)
ctx.sync()
ref = pipe.download()
ctx.timing_reset()
pipe.run(fed=False)
ctx.sync()
print('staged', timers(), flush=True)
for i in range(calls):
    for k in pipe.out:
        pipe.out[k].zero()
    ctx.timing_reset()
    t = time.perf_counter()
    err = None
    try:
        pipe.run(fed=True)
        ctx.sync()
    except _hip.HipError as exc:
        err = str(exc)[:120]
    wall = time.perf_counter() - t
    same = {k: bool(np.array_equal(pipe.out[k].download(), ref[k], equal_nan=True)) for k in pipe.out}
    print('fed call', i, 'wall ms', round(1e3 * wall, 2), timers(), 'identical', all(same.values()), same if not all(same.values()) else '',
          'reroutes', pipe.plan.info()['reroutes'], err or '', flush=True)
