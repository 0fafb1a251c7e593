import os, sys
os.environ['XH_ROUTE_VALIDATE_FIRST'] = '0'
sys.path.insert(0, '/root/repo')
import numpy as np
from types import SimpleNamespace as NS
from oracle import months as o_months, mrtm as o_mrtm
from xanthos_amd import _hip, synth
from xanthos_amd.routing import mrtm
w = synth.make_world(nrow=60, ncol=120, ncell=3000, n_basins=5, seed=3, outlet_frac=0.02)
st = NS(ngridrow=w.nrow, ngridcol=w.ncol)
um = mrtm.upstream_genmatrix(mrtm.upstream(w.coords, mrtm.downstream(w.coords, w.flow_dir, st), st))
rng = np.random.default_rng(9)
runoff = rng.gamma(2.0, 30.0, (w.ncell, 12))
ndays = o_months.set_month_arrays(12, 1972, 1972)[:, 2]
ref = o_mrtm.route_series(um.tocsr(), w.flow_dist, w.velocity, w.area, runoff, ndays, 2)
for name, flags in (('units', 4), ('exact', 256), ('lockstep', 8 | 256), ('reassoc', 128)):
    got = mrtm.route_series(um, w.flow_dist, w.velocity, w.area, runoff, ndays, 2, flags=flags)
    k = um.plan(_hip.get_context()).info()['last_tree_kernel']
    bad = [int((~((a == b) | (np.isnan(a) & np.isnan(b)))).sum()) for a, b in zip(got, ref)]
    rel = [float(np.nanmax(np.abs(a - b) / (np.abs(b) + 1e-9))) for a, b in zip(got, ref)]
    print(name, 'kernel', k, 'values differing', bad, 'max rel', rel, flush=True)
