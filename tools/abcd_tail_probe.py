"""Does k_abcd_tile's grid pay for a second round of workgroups?  67,420 cells / 32 per wave = 2,107 workgroups on 2,048 wave
slots (two 256-register waves per SIMD): time the ABCD simulation march for cell counts either side of 65,536.
python tools/abcd_tail_probe.py"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from xanthos_amd import _hip      # noqa: E402

ctx = _hip.get_context(0)
nm, spin, nb = 600, 120, 235
rng = np.random.default_rng(0)
nmax = 70000
pet = ctx.upload(rng.gamma(2.0, 40.0, (nmax, nm)))
pr = ctx.upload(rng.gamma(2.0, 45.0, (nmax, nm)))
tn = ctx.upload(rng.normal(5.0, 8.0, (nmax, nm)))
pars = ctx.upload(np.column_stack([rng.uniform(0.9, 0.99, nb), rng.uniform(0.2, 2.0, nb), rng.uniform(0.1, 0.9, nb),
                                   rng.uniform(0.1, 0.9, nb), rng.uniform(0.1, 0.9, nb)]))
out = [ctx.empty((nmax, nm)) for _ in range(3)]
for ncell in (32768, 49152, 61440, 65536, 65568, 67420, 69632):
    bidx = (np.arange(ncell) % nb).astype(np.int32)
    for rep in range(2):
        ctx.timing_reset()
        for _ in range(5):
            ctx.abcd(ncell, nm, spin, nb, bidx, bidx, nb, pars, pet, pr, tn, out[0], out[1], out[2])
        ctx.sync()
    ms, n = ctx.timing('abcd_sim')
    print('ncell {:6d}  workgroups {:5d}  abcd_sim {:.3f} ms  {:.2f} TB/s'.format(ncell, (ncell + 31) // 32, ms / n,
                                                                              ncell * nm * 48 / (ms / n * 1e-3) / 1e12), flush=True)
