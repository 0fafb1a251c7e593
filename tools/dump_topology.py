"""Write the routing topology of the synthetic world (and which cells can fire at dt = 3 h) for tests/plan_fuzz/plan_fuzz
--file: the planner's statistics and its invariants on the real 67,420-cell network without a GPU."""
import os
import sys
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
from xanthos_amd import synth
from xanthos_amd.pipeline import topology_from_world

ncell = int(sys.argv[2]) if len(sys.argv) > 2 else 67420
w = synth.make_world(ncell=ncell, n_basins=max(1, 235 * ncell // 67420))
um = topology_from_world(w)
if len(sys.argv) > 4:      # shard `rank` of `n_ranks` of the basin / network-closed partition (xanthos_amd.dist)
    from xanthos_amd.dist import make_shards, sub_world
    shard = make_shards(w, um, int(sys.argv[4]))[int(sys.argv[3])]
    w, um = sub_world(w, um, shard)
cap = ~((w.velocity / w.flow_dist) * 10800.0 <= 1.0 - 2.0 ** -20)
with open(sys.argv[1], 'wb') as f:
    f.write(np.int32(w.ncell).tobytes())
    f.write(np.int64(len(um.indices)).tobytes())
    f.write(np.ascontiguousarray(um.indptr, dtype=np.int64).tobytes())
    f.write(np.ascontiguousarray(um.indices, dtype=np.int32).tobytes())
    f.write(np.ascontiguousarray(um.sign, dtype=np.int8).tobytes())
    f.write(cap.astype(np.uint8).tobytes())
print(w.ncell, 'cells;', int(cap.sum()), 'can fire')
