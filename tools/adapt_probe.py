import sys, time, os
sys.path.insert(0, os.environ.get('GRAFT_REPO_ROOT', '/root/repo'))
import numpy as np
from xanthos_amd import _hip, synth
from xanthos_amd.pipeline import pipeline_from_world
ctx = _hip.get_context(0)
w = synth.make_world()
nm = 120
pipe = pipeline_from_world(ctx, w, nm, 1961, 60, 24)
ctx.synth_forcing(5, w.ncell, nm, ctx.upload(w.latitude), pipe.alloc_forcing(), nan_frac=0.001)
pipe.run(('pm', 'abcd'), fused=False); ctx.sync()
pipe.route_flags = _hip.XH_ROUTE_VALIDATE
for phase in range(3):
    if phase >= 1:
        q = pipe.out['q'].download()
        q2 = np.where(np.isnan(q), np.nan, 0.0)
        band = (np.arange(w.ncell) % 7) == (3 if phase == 1 else 5)
        q2[band] = np.abs(q[band]) * 40.0 + (1000.0 if phase == 2 else 0.0)
        pipe.out['q'].upload(q2)
    for rep in range(5):
        pipe.run_mrtm(); ctx.sync()
        print(phase, rep, pipe.plan.typed_info(), flush=True)
        time.sleep(0.25)
