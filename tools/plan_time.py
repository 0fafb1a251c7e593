import sys, time, os
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
import numpy as np
from xanthos_amd import _hip, synth
from xanthos_amd.pipeline import topology_from_world
ctx = _hip.get_context(0)
w = synth.make_world()
t = time.time(); um = topology_from_world(w); print('topology', round(time.time() - t, 4))
for rep in range(2):
    um._plans = {}
    t = time.time(); p = um.plan(ctx); ctx.sync(); print('um.plan', round(time.time() - t, 4))
t = time.time(); a = [ctx.empty((67420, 600)) for _ in range(6)]; ctx.sync(); print('6 x empty', round(time.time() - t, 4))
t = time.time(); d = ctx.upload(w.lct); ctx.sync(); print('upload lct', w.lct.nbytes / 1e6, 'MB', round(time.time() - t, 4))
