"""xh_comm_gather_rows with MORE THAN ONE rank on a one-GPU box.

RCCL refuses two ranks on one device, so the rank processes of these tests find ``tests/fake_rccl/librccl.so.1`` first on
their loader path: a test-only stand-in for the eight RCCL entry points csrc/xh_comm.hip binds at run time (host bounce
through /dev/shm, see fake_rccl.cpp).  Everything above it is the product: libxanthos_hip.so's grouped sends / receives,
the root's staging offsets and row scatters, ``dist.make_shards`` / ``sub_world`` / ``fill_shard_forcing`` and the
device pipeline.  The reference has no counterpart (its only parallelism is the joblib pool of abcd.py:369-382).

The rank processes are started by the test BEFORE they touch the GPU (plain child processes, no exec from a process that
holds a GPU context) and never import torch (which would bring the real librccl into the process).
"""
import json
import os
import subprocess
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), '..'))
FAKE = os.path.join(ROOT, 'tests', 'fake_rccl')

RANK_COMMON = r'''
import json, os, sys, time
root_dir, rank, nranks, root, workdir = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4]), sys.argv[5]
sys.path.insert(0, root_dir)
import numpy as np
from xanthos_amd import _hip as hip
assert 'torch' not in sys.modules
ctx = hip.get_context(0)
id_file = os.path.join(workdir, 'rccl_id')
if rank == root:
    uid = hip.comm_unique_id()
    assert uid[:4] == b'FAKE', 'the stand-in librccl was not the one loaded'
    with open(id_file + '.tmp', 'wb') as f:
        f.write(uid)
    os.rename(id_file + '.tmp', id_file)
else:
    t0 = time.time()
    while not os.path.exists(id_file):
        assert time.time() - t0 < 120, 'no RCCL id from the root'
        time.sleep(0.01)
    uid = open(id_file, 'rb').read()
comm = hip.Comm(ctx, nranks, rank, uid)
'''

# ---- 1. synthetic rows: every value encodes (variable, global row, column), so a misplaced block cannot hide
ROWS_CHILD = RANK_COMMON + r'''
counts = json.loads(sys.argv[6])
nvar, ncols = 6, 37
total = sum(counts)
perm = np.random.default_rng(7).permutation(total)              # rank-major row -> grid row
first = sum(counts[:rank])
mine = perm[first:first + counts[rank]]
val = lambda v, rows: v * 1e7 + rows[:, None] * 100.0 + np.arange(ncols)[None, :]
local = [ctx.upload(val(v, mine)) if counts[rank] else ctx.empty((0, ncols)) for v in range(nvar)]
for rep in range(2):                                            # twice: the staging area is reused, counters restart
    if rank == root:
        out = [ctx.empty((total, ncols)).zero() for _ in range(nvar)]
        comm.gather_rows(local, counts, ncols, perm=ctx.upload(perm, dtype=np.int64), out=out, root=root)
        ctx.sync()
        for v in range(nvar):
            assert np.array_equal(out[v].download(), val(v, np.arange(total))), (rep, v)
    else:
        comm.gather_rows(local, counts, ncols, root=root)
        ctx.sync()
comm.close()
print('RANK_OK', rank)
'''

# ---- 2. the sharded pipeline: each rank runs its shard of one world, the root compares with the unsharded run
PIPE_CHILD = RANK_COMMON + r'''
from xanthos_amd import synth
from xanthos_amd.dist import fill_shard_forcing, make_shards, sub_world
from xanthos_amd.pipeline import OUTPUTS, pipeline_from_world, topology_from_world
w = synth.make_world(nrow=60, ncol=120, ncell=3000, n_basins=9, seed=21)
um = topology_from_world(w)
nm, seed = 36, 5
shards = make_shards(w, um, nranks)
counts = [len(s.cells) for s in shards]
flags = hip.XH_ROUTE_NO_DATAFLOW       # several processes share this GPU: the all-units-resident kernels are not under test here
sw, sum_ = sub_world(w, um, shards[rank])
pipe = pipeline_from_world(ctx, sw, nm, 1971, 25, 6, um=sum_, route_flags=flags)
pipe.alloc_forcing()
fill_shard_forcing(ctx, w, shards[rank], pipe, seed, nan_frac=0.002)
pipe.run()
local = [pipe.out[k] for k in OUTPUTS]
if rank == root:
    perm = ctx.upload(np.concatenate([s.cells for s in shards]), dtype=np.int64)
    out = [ctx.empty((w.ncell, nm)) for _ in OUTPUTS]
    ctx.sync()
    comm.gather_rows(local, counts, nm, perm=perm, out=out, root=root)
    ctx.sync()
    whole = pipeline_from_world(ctx, w, nm, 1971, 25, 6, um=um, route_flags=flags)
    ctx.synth_forcing(seed, w.ncell, nm, ctx.upload(w.latitude), whole.alloc_forcing(), nan_frac=0.002)
    whole.run()
    ref = whole.download()
    for k, o in zip(OUTPUTS, out):
        got = o.download()
        assert np.array_equal(got, ref[k], equal_nan=True), k
    assert np.isnan(ref['q']).any() and np.isfinite(ref['avg']).any()
else:
    ctx.sync()
    comm.gather_rows(local, counts, nm, root=root)
    ctx.sync()
comm.close()
print('RANK_OK', rank, counts)
'''


def _fake_rccl():
    so = os.path.join(FAKE, 'librccl.so.1')
    if not os.path.isfile(so):                      # normally built by __graft_entry__.build()
        subprocess.run(['make', '-C', FAKE], check=True, capture_output=True)
    return so


def _run_ranks(tmp_path, source, nranks, root, extra=()):
    _fake_rccl()
    script = tmp_path / 'rank.py'
    script.write_text(source)
    env = dict(os.environ)
    env['LD_LIBRARY_PATH'] = FAKE + os.pathsep + env.get('LD_LIBRARY_PATH', '')
    procs = [subprocess.Popen([sys.executable, str(script), ROOT, str(r), str(nranks), str(root), str(tmp_path), *extra],
                              env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
             for r in range(nranks)]
    outs = []
    for p in procs:
        try:
            outs.append(p.communicate(timeout=420)[0])
        except subprocess.TimeoutExpired:
            for q in procs:
                q.kill()
            raise
    for r, (p, o) in enumerate(zip(procs, outs)):
        assert p.returncode == 0 and 'RANK_OK {}'.format(r) in o, 'rank {}:\n{}'.format(r, o[-3000:])
    leftovers = [f for f in os.listdir('/dev/shm') if f.startswith('xh_fake_rccl_')]
    assert not leftovers, leftovers                 # every message was received
    return outs


@pytest.mark.parametrize('nranks,root,counts', [
    (2, 0, [700, 300]),
    (2, 1, [300, 700]),
    (3, 1, [250, 0, 410]),          # an empty shard among the senders
    (3, 2, [0, 500, 77]),           # root last: every remote row lies before the root's own
    (3, 0, [0, 9, 130]),            # the root itself holds nothing
    (4, 2, [64, 1, 333, 5]),        # remote rows on both sides of the root's
])
def test_gather_rows_several_ranks(tmp_path, nranks, root, counts):
    """Six variables, unequal and empty shards, every root position: the gathered arrays equal the expected grid."""
    _run_ranks(tmp_path, ROWS_CHILD, nranks, root, extra=(json.dumps(counts),))


@pytest.mark.parametrize('nranks,root', [(2, 0), (3, 1)])
def test_sharded_pipeline_gathered_equals_unsharded(tmp_path, nranks, root):
    """PM -> ABCD -> MRTM on basin / network-closed shards in `nranks` processes, ONE xh_comm_gather_rows of the six
    outputs to `root`: bit-identical (NaN runoff included) to the unsharded run on the same device."""
    _run_ranks(tmp_path, PIPE_CHILD, nranks, root)


def test_bench_gpus_2_starts_its_own_ranks(tmp_path):
    """``python bench.py --gpus 2 --steps 3 --warmup 1`` with NO launcher and no RANK in the environment: the script starts
    its two rank processes itself (before anything touches the GPU), both on this box's one GPU (XH_BENCH_ONE_DEVICE=1, the
    process group on gloo), the write-out gather through libxanthos_hip.so's grouped sends / receives with the test-only
    stand-in first on the loader path; rank 0's JSON line is relayed, says two ranks were seen and which gather ran, and the
    gathered arrays equal the same world run unsharded (--check-gather), all six outputs, bit for bit.  Routing runs with
    one workgroup per network (--route-flags 4): two processes cannot both keep every dataflow unit resident on one GPU."""
    _fake_rccl()
    env = dict(os.environ)
    for k in ('RANK', 'WORLD_SIZE', 'LOCAL_RANK', 'MASTER_ADDR', 'MASTER_PORT'):
        env.pop(k, None)
    # (the rank processes import torch for the gloo process group, and PyTorch's bundled RCCL has the soname librccl.so.1:
    # the stand-in is named by its path)
    env.update({'XH_BENCH_ONE_DEVICE': '1', 'XH_BENCH_BACKEND': 'gloo', 'XH_FAKE_RCCL_DIR': str(tmp_path),
                'XH_RCCL_LIBRARY': _fake_rccl()})
    out = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py'), '--gpus', '2', '--steps', '3', '--warmup', '1',
                          '--route-flags', '4', '--check-gather', '--no-replica-figure'],
                         env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=1500)
    assert out.returncode == 0, out.stderr[-4000:]
    lines = [ln for ln in out.stdout.splitlines() if ln.startswith('{')]
    assert len(lines) == 1, out.stdout[-2000:]
    res = json.loads(lines[0])
    assert res['n_gpus'] == 2 and res['n_ranks_seen'] == 2 and res['steps'] == 3 and res['scaling'] == 'strong'
    g = res['gather']
    assert g['kind'] == 'rccl' and 'stand-in' in g['library'], g
    assert g['equals_unsharded'] is True and all(g['equals_unsharded_by_output'].values()), g
    assert len(g['rows_per_rank']) == 2 and sum(g['rows_per_rank']) == 67420
    assert 'exposed_ms' in g and res['value'] > 0
    assert not [f for f in os.listdir(str(tmp_path)) if f.startswith('xh_fake_rccl_')]      # every message was received


# ---- run_model() on N ranks: the product's own multi-GPU path
RUN_MODEL_PARENT = r'''
import os, sys
sys.path.insert(0, sys.argv[1])
from xanthos_amd import run_model
assert 'torch' not in sys.modules
# (this process never touches a GPU: with gpus > 1 it only starts the rank processes and waits for them)
res = run_model(sys.argv[2], gpus=int(sys.argv[3]))
assert (res is None) == (int(sys.argv[3]) > 1)
assert 'torch' not in sys.modules
print('PARENT_OK')
'''


@pytest.mark.parametrize('nranks,form', [(2, 'exact'), (3, 'default')])
def test_run_model_on_n_ranks_writes_the_same_files(tmp_path, nranks, form):
    """``run_model(ini, gpus=N)``: the launcher starts N rank processes (all on this box's one GPU: XH_ONE_DEVICE=1), each maps
    its rows of the forcing files, runs the pipeline on its basins, the six outputs travel to rank 0 through the library's
    own gather (grouped ncclSend / ncclRecv; the RCCL stand-in of tests/fake_rccl bounces them through files) and rank 0
    writes -- post-processors and runoff aggregations included.  With the bit-exact routing kernels (``routing_form = exact``
    in the ini) every .npy / .csv of the output folder is byte-identical to the one-rank run's.  With the library's default
    (the reassociated form, whose sums follow the chains of the partition -- and a shard's partition is not the whole
    world's) every file but the channel flow is byte-identical and the channel flow agrees within 1e-9.  No torch anywhere:
    the ranks meet over launch.SocketGroup."""
    from xanthos_amd import synth
    w = synth.make_world(nrow=36, ncol=72, ncell=900, n_basins=7, seed=33)
    f = synth.make_forcing(w, 36)
    outs = {}
    for tag, n in (('one', 1), ('many', nranks)):
        root = str(tmp_path / tag)
        os.makedirs(root)
        ini = synth.write_example(root, w, f, 1971, 1973, runoff_spinup=25, routing_spinup=6, post=True, aggregates=True,
                                  output_vars=('pet', 'aet', 'q', 'soilmoisture', 'avgchflow'),
                                  output_format=4 if nranks == 2 else 1)          # .npy with two ranks, .csv with three
        if form == 'exact':
            text = open(ini).read()
            assert 'routing_spinup' in text
            open(ini, 'w').write(text.replace('routing_spinup', 'routing_form = exact\n    routing_spinup', 1))
        script = tmp_path / (tag + '.py')
        script.write_text(RUN_MODEL_PARENT)
        env = dict(os.environ)
        env.update({'XH_ONE_DEVICE': '1', 'XH_RCCL_LIBRARY': _fake_rccl(), 'XH_FAKE_RCCL_DIR': str(tmp_path)})
        for k in ('RANK', 'WORLD_SIZE', 'XH_ROUTE_REASSOC'):      # (the library's default unless the ini says otherwise)
            env.pop(k, None)
        r = subprocess.run([sys.executable, str(script), ROOT, ini, str(n)], env=env, capture_output=True, text=True, timeout=600)
        assert r.returncode == 0 and 'PARENT_OK' in r.stdout, r.stdout[-3000:] + r.stderr[-3000:]
        if n > 1:
            assert 'gather (rccl)' in r.stdout, r.stdout[-3000:]          # the library's gather, not the host fall-back
            assert r.stdout.count('of 900 cells on this rank') == n
        outs[tag] = os.path.join(root, 'output', 'pm_abcd_mrtm_synth')
    files = sorted(x for x in os.listdir(outs['one']) if x.endswith(('.npy', '.csv')))
    assert len(files) >= 6 and files == sorted(x for x in os.listdir(outs['many']) if x.endswith(('.npy', '.csv'))), files
    n_close = 0
    for name in files:
        one, many = (open(os.path.join(outs[t], name), 'rb').read() for t in ('one', 'many'))
        if one == many:
            continue
        assert form == 'default' and 'avgchflow' in name, name          # only the routed variable, only in the reassociated form
        a, b = (np.genfromtxt(os.path.join(outs[t], name), delimiter=',', skip_header=1) for t in ('one', 'many'))
        assert a.shape == b.shape and np.array_equal(np.isnan(a), np.isnan(b)), name
        m = ~np.isnan(a)
        assert (np.abs(a[m] - b[m]) <= 1e-9 * np.abs(a[m]) + 1e-9).all(), (name, float(np.abs(a[m] - b[m]).max()))
        n_close += 1
    assert n_close <= 1
    assert not [x for x in os.listdir(str(tmp_path)) if x.startswith('xh_fake_rccl_')]      # every message was received
