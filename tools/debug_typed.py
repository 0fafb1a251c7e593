"""Debugging aid: replay one case of tools/fuzz_routing.py and say WHICH cells differ from the oracle and where they sit
in the typed partition (run on the GPU box).  Usage: python tools/debug_typed.py <seed> <case>"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
os.environ['XH_FLOW_DUMP'] = '/tmp/flow_dump.bin'
os.environ['XH_FLOW_CHECK'] = '1'
os.environ['XH_FLOW_STATS'] = '1'
import tools.fuzz_routing as fz
from oracle import mrtm as o_mrtm
from xanthos_amd import _hip, synth
from xanthos_amd.routing import mrtm
from types import SimpleNamespace as NS

seed, case = int(sys.argv[1]), int(sys.argv[2])
rng = np.random.default_rng(seed)
for k in range(case + 1):
    c = fz.gen_case(rng)
s = NS(um=c.um, L=c.L, v=c.v, area=c.w.area, q=c.q, ndays=c.ndays, spin=c.spin, S0=c.S0, dt=c.dt)
print('case', case, 'cells', c.ncell, 'months', c.nm, 'spin', c.spin, 'dt', c.dt, 'S0', c.S0 is not None)
ref = o_mrtm.route_series(s.um.tocsr(), s.L, s.v, s.area, s.q, s.ndays, s.spin, S0=s.S0, dt=s.dt)
got = mrtm.route_series(s.um, s.L, s.v, s.area, s.q, s.ndays, s.spin, S0=s.S0, dt=s.dt, flags=0)      # the FIRST call on this plan
plan = s.um.plan(_hip.get_context(0))
raw = np.fromfile('/tmp/flow_dump.bin', dtype=np.int32)
n = raw[0]
ds, piece, unit, hgt, up = (raw[1 + i * n: 1 + (i + 1) * n] for i in range(5))
print('typed', plan.typed_info(), 'reroutes', plan.info()['reroutes'])
st = plan.stats()
if st is not None:
    sh = (st[:, 3] & np.uint64(255)).astype(int)
    print('units', len(st), 'plain', int(((sh & 64) != 0).sum()), 'units whose guard variables are set', np.nonzero(sh & 128)[0][:20], 'of them plain', int((((sh & 128) != 0) & ((sh & 64) != 0)).sum()), 'unit of 2920 guard', sh[unit[2920]] if 'unit' in dir() else '')
cap = ~((s.v / s.L) * s.dt <= 1 - 2.0 ** -20)
bad = np.zeros(n, bool)
for name, a, b in zip(('chs', 'avg', 'F_end', 'S_end?'), got, ref):
    m = ~((a == b) | (np.isnan(a) & np.isnan(b)))
    print(name, 'mismatching values', int(m.sum()), 'first months', np.unique(np.nonzero(m)[1])[:10] if m.ndim == 2 else '')
    bad |= m.reshape(n, -1).any(axis=1)
cells = np.nonzero(bad)[0]
print('bad cells', len(cells), 'of which plain', int(((up[cells] & 0x200) != 0).sum()), 'capable', int(cap[cells].sum()))
# are the bad cells closed downstream (errors propagate) -- find the most upstream ones
badset = set(cells.tolist())
ups = [c for c in cells if not any((ds == c) & bad)]
print('most upstream bad cells', len(ups))
for c in ups[:12]:
    prod = np.nonzero(ds == c)[0]
    print(' cell', c, 'unit', unit[c], 'shape %x' % up[c], 'hgt', hgt[c], 'cap', bool(cap[c]), 'nan q', bool(np.isnan(s.q[c]).any()),
          '| producers', [(int(p), int(unit[p]), '%x' % up[p], bool(cap[p]), bool(np.isnan(s.q[p]).any())) for p in prod],
          '| ds', int(ds[c]), int(unit[ds[c]]) if ds[c] >= 0 else None)
    print('   got chs', got[0][c, :4], 'ref', ref[0][c, :4])
    print('   got avg', got[1][c, :4], 'ref', ref[1][c, :4])
