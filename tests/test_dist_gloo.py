"""Multi-process CPU tests (world_size 2 and 3) of the basin sharding, the launcher, the process group -- the package's own TCP
rendezvous and torch.distributed's gloo backend behind the same interface -- and the fall-back of the single output gather."""
import os
import socket
import subprocess
import sys
import time

import numpy as np

from xanthos_amd import synth
from xanthos_amd.dist import make_shards, shard_components, sub_matrix
from xanthos_amd.pipeline import topology_from_world

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), '..'))

WORKER = r'''
import os, sys
sys.path.insert(0, sys.argv[1])
import numpy as np
from xanthos_amd import synth
from xanthos_amd.dist import make_shards, host_gather
from xanthos_amd.pipeline import topology_from_world
if sys.argv[2] == 'gloo':              # torch.distributed (gloo) behind the same small interface: bench.py's adapter
    import torch, torch.distributed as dist
    dist.init_process_group(backend='gloo', rank=int(os.environ['RANK']), world_size=int(os.environ['WORLD_SIZE']))
    import bench
    group = bench.TorchGroup(dist, torch, 'gloo')
else:                                  # the package's own TCP rendezvous (no torch anywhere in this process)
    from xanthos_amd import launch
    group = launch.current_group()
    assert 'torch' not in sys.modules
rank = group.rank
w = synth.make_world(nrow=36, ncol=72, ncell=900, n_basins=7, seed=33)
um = topology_from_world(w)
shards = make_shards(w, um, group.size)
mine = shards[rank].cells
# "outputs" of this rank: value encodes (variable, global cell, month) so the reassembly can be checked exactly
nvar, nm = 3, 5
local = np.stack([v * 1e6 + mine[:, None] * 10.0 + np.arange(nm)[None, :] for v in range(nvar)])
assert group.allreduce(len(mine), 'sum') == w.ncell and group.allreduce(rank, 'max') == group.size - 1
assert group.bcast(b'id' * 64 if rank == 0 else None, src=0) == b'id' * 64
out = host_gather(local, shards, w.ncell, group)
if rank == 0:
    want = np.stack([v * 1e6 + np.arange(w.ncell)[:, None] * 10.0 + np.arange(nm)[None, :] for v in range(nvar)])
    assert out.shape == (nvar, w.ncell, nm)
    assert np.array_equal(out, want)
    print('GATHER_OK', [len(s.cells) for s in shards])
else:
    assert out is None
group.barrier()
'''


def test_shards_are_closed_and_balanced():
    w = synth.make_world(nrow=36, ncol=72, ncell=900, n_basins=7, seed=33)
    um = topology_from_world(w)
    labels = shard_components(w.basin_ids, um)
    # every basin and every flow edge stays inside one component
    for b in np.unique(w.basin_ids):
        assert len(np.unique(labels[w.basin_ids == b])) == 1
    rows = np.repeat(np.arange(w.ncell), np.diff(um.indptr))
    assert (labels[rows] == labels[um.indices]).all()
    for n in (2, 4):
        shards = make_shards(w, um, n)
        allc = np.concatenate([s.cells for s in shards])
        assert len(allc) == w.ncell and len(np.unique(allc)) == w.ncell
        sizes = np.array([len(s.cells) for s in shards])
        assert sizes.max() - sizes.min() <= np.bincount(labels).max()          # LPT bound
        for s in shards:
            sub = sub_matrix(um, s.cells)                                        # raises if an edge leaves the shard
            assert sub.shape[0] == len(s.cells) and len(sub.indices) == np.diff(um.indptr)[s.cells].sum()


import pytest


@pytest.mark.parametrize('kind', ['socket', 'gloo'])
def test_host_gather_two_ranks(tmp_path, kind):
    """The fall-back of the write-out gather (rows through the process group) and the group's small collectives, with two
    rank processes on the CPU: through the package's own TCP rendezvous (launch.SocketGroup: what run_model() uses, no
    torch in the process) and through torch.distributed's gloo backend behind bench.py's adapter."""
    script = tmp_path / 'worker.py'
    script.write_text(WORKER)
    with socket.socket() as s:
        s.bind(('127.0.0.1', 0))
        port = s.getsockname()[1]
    procs = []
    for rank in range(2):
        env = dict(os.environ, RANK=str(rank), WORLD_SIZE='2', MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port))
        procs.append(subprocess.Popen([sys.executable, str(script), ROOT, kind], env=env, stdout=subprocess.PIPE,
                                      stderr=subprocess.STDOUT, text=True))
    outs = [p.communicate(timeout=300)[0] for p in procs]
    assert all(p.returncode == 0 for p in procs), outs
    assert 'GATHER_OK' in outs[0]


def test_launcher_starts_ranks_and_relays_exit_codes(tmp_path):
    """launch.spawn: N plain child processes with RANK / LOCAL_RANK / WORLD_SIZE / MASTER_ADDR / MASTER_PORT set, a group
    that really connects them (three ranks: a max over the ranks, an object from rank 2), the worst exit code returned."""
    script = tmp_path / 'rank.py'
    script.write_text("""
import os, sys
sys.path.insert(0, sys.argv[1])
from xanthos_amd import launch
g = launch.current_group()
rank, local, world = launch.env_world()
assert (g.rank, g.size) == (rank, world) == (int(os.environ['RANK']), 3) and local == (0 if sys.argv[2] == '1' else rank)
assert g.allreduce(rank * 10, 'max') == 20 and g.bcast('x' if rank == 2 else None, src=2) == 'x'
got = g.gather({'rank': rank}, root=0)
assert (got == [{'rank': 0}, {'rank': 1}, {'rank': 2}]) if rank == 0 else got is None
g.barrier()
print('RANK_OK', rank, flush=True)
launch.close_group()
sys.exit(int(sys.argv[3]) if rank == 1 else 0)
""")
    code = ("import sys; sys.path.insert(0, %r); from xanthos_amd import launch; "
            "sys.exit(launch.spawn(3, [%r, %r, sys.argv[1], sys.argv[2]], one_device=sys.argv[1] == '1'))" % (ROOT, str(script), ROOT))
    for one_device, rc_want in (('0', 0), ('1', 7)):
        out = subprocess.run([sys.executable, '-c', code, one_device, str(rc_want)], capture_output=True, text=True, timeout=120)
        assert out.returncode == rc_want, out.stdout + out.stderr
        assert out.stdout.count('RANK_OK') == 3 and '[rank 2] RANK_OK 2' in out.stdout, out.stdout


CALIB_WORKER = r'''
import os, sys
sys.path.insert(0, sys.argv[1])
from types import SimpleNamespace as NS
import numpy as np
from xanthos_amd import launch
from xanthos_amd.calibrate import calibrate_abcd as cal
group = launch.current_group()
rank = group.rank
rng = np.random.default_rng(2)
basin_ids = rng.integers(1, 8, 400)                 # 7 basins of different sizes; basin 8 requested but empty
nm = 30
settings = NS(set_calibrate=0, obs_unit='km3_per_mth', cal_basins=['1-8'], nmonths=nm, runoff_spinup=25,
              calib_out_dir=sys.argv[2], device=0)
obs = np.concatenate([np.stack([np.full(nm, b), np.arange(nm) + b], axis=1) for b in range(1, 9)])
data = NS(basin_ids=basin_ids, area=np.ones(400), precip=np.ones((400, nm)), tmin=np.ones((400, nm)), cal_obs=obs)
seen = []

def fake_local(mine, settings, data, pet, seed, popsize, nmembers):     # stands in for the GPU search of this rank
    seen.extend(mine)
    rows = np.array([[b + 0.1, b + 0.2, b + 0.3, b + 0.4, b + 0.5, 1.0 / b, 75 * b, b] for b in mine]).reshape(-1, 8)
    return rows, {}
cal._calibrate_local = fake_local
res = cal.calibrate_all(settings, data, np.ones((400, nm)), seed=1, group=group)
sizes = np.array([(basin_ids == b).sum() for b in range(1, 8)])
owner = cal.assign_basins(sizes * nm, 2)
assert seen == [b for b, r in zip(range(1, 8), owner) if r == rank] and 0 < len(seen) < 7
if rank == 0:
    assert sorted(res) == list(range(1, 8))
    for b, (x, kge) in res.items():
        assert np.array_equal(x, b + np.array([0.1, 0.2, 0.3, 0.4, 0.5])) and kge == 1 - 1.0 / b
        assert np.load(os.path.join(sys.argv[2], 'kge_result_basin_%d.npy' % b))[0] == kge
        assert np.array_equal(np.load(os.path.join(sys.argv[2], 'abcdm_parameters_basin_%d.npy' % b))[0], x)
    print('CALIB_FANOUT_OK', owner.tolist())
else:
    assert res == {}
group.barrier()
assert 'torch' not in sys.modules
'''


def _run_two_ranks(tmp_path, text, *extra):
    script = tmp_path / 'worker.py'
    script.write_text(text)
    with socket.socket() as s:
        s.bind(('127.0.0.1', 0))
        port = s.getsockname()[1]
    procs = []
    for rank in range(2):
        env = dict(os.environ, RANK=str(rank), WORLD_SIZE='2', MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port))
        procs.append(subprocess.Popen([sys.executable, str(script), ROOT] + list(extra), env=env,
                                      stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True))
    outs = [p.communicate(timeout=300)[0] for p in procs]
    assert all(p.returncode == 0 for p in procs), outs
    return outs


def test_calibration_fanout_two_ranks(tmp_path):
    """calibrate_all over 2 ranks: basins dealt by size, each rank searches only its share (the GPU search is
    replaced by a stand-in), ONE collective brings the [n_basins, n_par + 3] table to rank 0, which writes the files."""
    out = tmp_path / 'calib_out'
    outs = _run_two_ranks(tmp_path, CALIB_WORKER, str(out))
    assert 'CALIB_FANOUT_OK' in outs[0]


def test_group_takes_no_strangers_and_nothing_pickled(tmp_path):
    """the rendezvous never unpickles, a peer has to present the job's token in a fixed-size hello, and the
    raw gather works towards any root.  Three ranks with a token; while rank 0 listens a STRANGER connects first and sends a
    pickle (it must be dropped, costing the job nothing), then a hello with the wrong token (dropped too)."""
    script = tmp_path / 'rank.py'
    script.write_text("""
import os, pickle, socket, struct, sys, time
sys.path.insert(0, sys.argv[1])
from xanthos_amd import launch
assert 'pickle' not in open(launch.__file__).read().replace('unpickled', '').replace('no pickle', '')
rank, _, world = launch.env_world()
port = int(os.environ['MASTER_PORT'])
if rank == 1:      # plays the stranger before joining properly
    for payload in (struct.pack('<Q', 40) + pickle.dumps(os.getcwd)[:40], launch._HELLO.pack(b'XHG1', 2, b'wrong'.ljust(64, b'\\0'))):
        for _ in range(200):
            try:
                s = socket.create_connection(('127.0.0.1', port), timeout=1.0)
                break
            except OSError:
                time.sleep(0.05)
        s.sendall(payload)
        time.sleep(0.2)
        s.close()
g = launch.current_group()
got = g.gather(bytes([rank]) * (3 + rank), root=1, raw=True)
assert (got == [b'\\x00' * 3, b'\\x01' * 4, b'\\x02' * 5]) if rank == 1 else got is None, got
import numpy as np
tab = g.gather(np.full((2, 3), float(rank)), root=0)
assert rank != 0 or [float(t[0, 0]) for t in tab] == [0.0, 1.0, 2.0]
assert g.allreduce(rank, 'sum') == 3
print('GROUP_OK', rank, flush=True)
launch.close_group()
""")
    code = ("import sys; sys.path.insert(0, %r); from xanthos_amd import launch; "
            "sys.exit(launch.spawn(3, [%r, %r], one_device=True))" % (ROOT, str(script), ROOT))
    out = subprocess.run([sys.executable, '-c', code], capture_output=True, text=True, timeout=180)
    assert out.returncode == 0 and out.stdout.count('GROUP_OK') == 3, out.stdout + out.stderr


def test_launcher_ends_the_job_when_a_rank_dies(tmp_path):
    """a rank that dies while the others sit in a collective must not leave run_model(gpus=N) hanging: the
    launcher supervises all ranks, gives the survivors a grace period and terminates them, and returns the first failure it
    notices (the dead rank's 9 -- or the 1 of a peer whose collective broke on the closed connection within the same poll)."""
    script = tmp_path / 'rank.py'
    script.write_text("""
import os, sys, time
sys.path.insert(0, sys.argv[1])
from xanthos_amd import launch
g = launch.current_group()
if g.rank == 1:
    os._exit(9)                      # dies after the rendezvous
g.barrier()                          # the others wait for it here
time.sleep(600)
""")
    code = ("import sys; sys.path.insert(0, %r); from xanthos_amd import launch; "
            "sys.exit(launch.spawn(3, [%r, %r], one_device=True))" % (ROOT, str(script), ROOT))
    t0 = time.time()
    out = subprocess.run([sys.executable, '-c', code], capture_output=True, text=True, timeout=120,
                         env=dict(os.environ, XH_SPAWN_GRACE='2'))
    assert out.returncode in (9, 1) and time.time() - t0 < 60, (out.returncode, out.stdout + out.stderr)
