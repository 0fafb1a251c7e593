"""What the fillers of a fed step cost the routing units ("the measured interference, cycles per
sub-step with / without the mates").  XH_FLOW_STATS=1 makes every unit of k_mrtm_wave record its shader cycles; the same
world is routed stage by stage (the units have their SIMDs to themselves) and fed (PM and ABCD waves of the remaining months
beside them for the first part of the run), steady state of the plan in both cases.
python tools/feed_interference.py"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ['XH_FLOW_STATS'] = '1'
from xanthos_amd import _hip, synth                      # noqa: E402
from xanthos_amd.pipeline import pipeline_from_world    # noqa: E402

ctx = _hip.get_context(0)
w = synth.make_world()
pipe = pipeline_from_world(ctx, w, 600, 1961, 120, 120)
ctx.synth_forcing(3, w.ncell, 600, ctx.upload(w.latitude), pipe.alloc_forcing(), nan_frac=0.001)
nsub = sum(int(d) * 8 for d in pipe.ndays) + sum(int(d) * 8 for d in pipe.ndays[:120])
for _ in range(6):                                       # the plan settles into its selective plain form
    pipe.run(fed=False)
    ctx.sync()


def one(fed):
    rows = []
    for _ in range(3):
        ctx.timing_reset()
        pipe.run(fed=fed)
        ctx.sync()
        st = pipe.plan.stats().astype(np.float64)
        rows.append((ctx.timing('mrtm_route')[0], st))
    ms = np.median([r[0] for r in rows])
    st = rows[len(rows) // 2][1]
    return ms, st


for name, fed in (('stage by stage', False), ('fed', True), ('stage by stage', False), ('fed', True)):
    ms, st = one(fed)
    shape = pipe.plan.stats()[:, 3]
    linked = (shape & np.uint64(48)) != 0
    total, waits = st[:, 1], st[:, 4] + st[:, 5]
    busy = (total - waits) / nsub
    clock = total / (st[:, 2] / 100e6) / 1e9                 # shader cycles per second of the 100 MHz real-time counter
    wall = st[:, 2] / 1e5                                    # ms from a unit's first to its last instruction
    print('{:15s} mrtm_route {:.2f} ms | cycles per sub-step outside waits, stream-linked units: median {:.0f}  p90 {:.0f}  max {:.0f}'
          ' | all units: median {:.0f}  max {:.0f} | waits per sub-step: median {:.0f} | shader clock GHz median {:.3f} min {:.3f}'
          ' | unit wall ms median {:.2f} max {:.2f}'.format(
              name, ms, np.median(busy[linked]), np.percentile(busy[linked], 90), busy[linked].max(), np.median(busy), busy.max(),
              np.median(waits[linked] / nsub), np.median(clock), clock.min(), np.median(wall), wall.max()), flush=True)
