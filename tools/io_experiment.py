"""File <-> HBM variants, timed on the GPU box (324 MB float64 arrays as in run_model() at the full grid):
upload: xh_upload_file | np.load(mmap_mode='r') + xh_memcpy_h2d | np.load + xh_memcpy_h2d
download: xh_download_file | np.lib.format.open_memmap + xh_memcpy_d2h | xh_memcpy_d2h + np.save"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
from xanthos_amd import _hip
from xanthos_amd.pipeline import file_range_of

ctx = _hip.get_context(0)
root = sys.argv[1] if len(sys.argv) > 1 else '/tmp/xh_io_exp'
os.makedirs(root, exist_ok=True)
shape = (67420, 600)
a = np.random.default_rng(0).standard_normal(shape)
paths = [os.path.join(root, 'in%d.npy' % k) for k in range(4)]
for p in paths:
    np.save(p, a)
d = ctx.empty(shape)
ctx.sync()
for rep in range(2):
    t = time.time()
    for p in paths:
        mm = np.load(p, mmap_mode='r')
        fn, off = file_range_of(mm)
        ctx.upload_file(d, fn, off, mm.nbytes)
    ctx.sync(); t1 = time.time() - t
    ok1 = np.array_equal(d.download(), a)
    t = time.time()
    for p in paths:
        d.upload(np.load(p, mmap_mode='r'))
    ctx.sync(); t2 = time.time() - t
    t = time.time()
    for p in paths:
        d.upload(np.load(p))
    ctx.sync(); t3 = time.time() - t
    gb = 4 * a.nbytes / 1e9
    print('upload %d: xh_upload_file %.3f s (%.1f GB/s, ok %s) | memmap + h2d %.3f s (%.1f GB/s) | np.load + h2d %.3f s (%.1f GB/s)' % (
        rep, t1, gb / t1, ok1, t2, gb / t2, t3, gb / t3))
outs = [os.path.join(root, 'out%d.npy' % k) for k in range(2)]
for rep in range(2):
    for p in outs:
        if os.path.exists(p): os.remove(p)
    t = time.time(); ctx.save_npy_many([(p, d) for p in outs]); t1 = time.time() - t
    ok1 = all(np.array_equal(np.load(p), a) for p in outs)
    for p in outs: os.remove(p)
    t = time.time()
    for p in outs:
        mm = np.lib.format.open_memmap(p, mode='w+', dtype=np.float64, shape=shape)
        d.download(out=mm)
        mm.flush(); del mm
    t2 = time.time() - t
    ok2 = all(np.array_equal(np.load(p), a) for p in outs)
    for p in outs: os.remove(p)
    t = time.time()
    for p in outs:
        np.save(p, d.download())
    t3 = time.time() - t
    gb = 2 * a.nbytes / 1e9
    print('download %d: xh_download_files %.3f s (%.1f GB/s, ok %s) | open_memmap + d2h %.3f s (%.1f GB/s, ok %s) | d2h + np.save %.3f s (%.1f GB/s)' % (
        rep, t1, gb / t1, ok1, t2, gb / t2, ok2, t3, gb / t3))
import shutil; shutil.rmtree(root, ignore_errors=True)
