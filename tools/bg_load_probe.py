"""The routing call beside another context's PM / ABCD kernels (tests/test_gpu_fullsize.py::
test_config3_twenty_repetitions_and_background_load without the comparisons): which kernel routed each call, how long it took."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
from xanthos_amd import _hip, synth
from xanthos_amd.pipeline import pipeline_from_world

ctx = _hip.get_context(0)
w = synth.make_world()
pipe = pipeline_from_world(ctx, w, 600, 1961, 120, 120)
f = pipe.alloc_forcing()
ctx.synth_forcing(1, pipe.ncell, pipe.nmonths, ctx.upload(w.latitude), f, nan_frac=0.001)
pipe.run(('pm', 'abcd'), fed=False)
pipe.run_mrtm()
ctx.sync()
other = _hip.Context(0)
bg = pipeline_from_world(other, w, 120, 1961, 30, 0)
other.synth_forcing(4, w.ncell, 120, other.upload(w.latitude), bg.alloc_forcing(), nan_frac=0.0)
nbg = int(sys.argv[1]) if len(sys.argv) > 1 else 40
bad = 0
for rep in range(int(sys.argv[2]) if len(sys.argv) > 2 else 20):
    if rep % 5 == 4:
        for _ in range(nbg):
            bg.run(('pm', 'abcd'))
    t0 = time.time()
    pipe.run_mrtm()
    ctx.sync()
    dt = time.time() - t0
    k = pipe.plan.info()['last_tree_kernel']
    bad += k != 4
    print('rep %2d: %8.1f ms, routed by kernel %d%s' % (rep, dt * 1e3, k, '  (beside the background load)' if rep % 5 == 4 else ''), flush=True)
    other.sync()
print('calls not routed by k_mrtm_rsum:', bad)
