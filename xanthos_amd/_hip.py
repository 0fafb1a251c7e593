"""ctypes binding of libxanthos_hip.so (include/xanthos_hip.h).

This is the only place the shared library is loaded.  There is NO CPU fallback: if the library is missing, or no
MI355X is visible, the calls raise :class:`HipUnavailable` -- the product path never routes through numpy.
"""
import ctypes
import os
from ctypes import POINTER, Structure, byref, c_char_p, c_double, c_int, c_int8, c_int32, c_int64, c_size_t, \
    c_uint64, c_void_p

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
# XH_LIBRARY: another build of the library (A/B experiments: `make -C xanthos_amd/csrc exp EXPNAME=... EXPFLAGS=...`)
LIB_PATH = os.environ.get('XH_LIBRARY') or os.path.join(_HERE, 'libxanthos_hip.so')

XH_ROUTE_DEFAULT, XH_ROUTE_FORCE_FALLBACK, XH_ROUTE_ATOMIC, XH_ROUTE_NO_DATAFLOW, XH_ROUTE_NO_SKEW = 0, 1, 2, 4, 8
XH_ROUTE_TEST_FAULT, XH_ROUTE_VALIDATE = 16, 32
XH_ROUTE_REASSOC, XH_ROUTE_EXACT = 128, 256      # reassociated (tolerance) form of the routing kernel / the bit-exact kernels
XH_ROUTE_NO_PLAIN = 0x4000                       # (xh_common.h, what a guard trip re-routes with) pairs of sums in every unit: not the prepared plan


class HipUnavailable(RuntimeError):
    """The HIP library or device is not usable; nothing falls back to the CPU."""


class HipError(RuntimeError):
    """A libxanthos_hip call returned an error code."""


class PmTables(Structure):
    _fields_ = [('nlcs', c_int32)] + [(n, POINTER(c_double)) for n in (
        'cL', 'beta', 'rslimit', 'Tminopen', 'Tminclose', 'VPDclose', 'VPDopen', 'RBLmin', 'RBLmax', 'rc', 'emiss',
        'alpha', 'lai', 'laimin', 'laimax')]


class FusedArgs(Structure):
    """xh_fused_args (include/xanthos_hip.h)."""
    _fields_ = ([('ncell', c_int64), ('nmonths', c_int32), ('start_year', c_int32), ('pm', POINTER(PmTables)),
                 ('n_lc_years', c_int32), ('h_lc_years', c_void_p), ('water_idx', c_int32), ('snow_idx', c_int32)] +
                [(n, c_void_p) for n in ('d_tas', 'd_tmin', 'd_rhs', 'd_wind', 'd_rsds', 'd_rlds', 'd_tairprev', 'd_lct',
                                         'd_elev')] +
                [('abcd_spinup', c_int32), ('n_groups', c_int32), ('h_basin_index', c_void_p), ('h_par_index', c_void_p),
                 ('npar_rows', c_int64), ('d_pars', c_void_p), ('d_precip', c_void_p), ('d_abcd_tmin', c_void_p),
                 ('plan', c_void_p), ('routing_spinup', c_int32), ('h_ndays', c_void_p), ('dt', c_double),
                 ('d_flow_dist', c_void_p), ('d_velocity', c_void_p), ('d_area', c_void_p), ('d_S0', c_void_p),
                 ('route_flags', c_int32)] +
                [(n, c_void_p) for n in ('d_pet', 'd_aet', 'd_q', 'd_sav', 'd_chstorage', 'd_avgchflow')] +
                [('block_months', c_int32), ('mode', c_int32)])


_P = c_void_p
# name -> (restype, argtypes); mirrors include/xanthos_hip.h one to one
SIGNATURES = {
    'xh_abi_version': (c_int, []),
    'xh_device_count': (c_int, [POINTER(c_int)]),
    'xh_ctx_create': (c_int, [c_int, POINTER(c_void_p)]),
    'xh_ctx_destroy': (None, [_P]),
    'xh_last_error': (c_char_p, [_P]),
    'xh_device_name': (c_int, [_P, ctypes.c_char_p, c_size_t]),
    'xh_malloc': (c_int, [_P, c_size_t, POINTER(c_void_p)]),
    'xh_free': (c_int, [_P, _P]),
    'xh_memcpy_h2d': (c_int, [_P, _P, _P, c_size_t]),
    'xh_memcpy_d2h': (c_int, [_P, _P, _P, c_size_t]),
    'xh_memcpy_d2d': (c_int, [_P, _P, _P, c_size_t]),
    'xh_host_alloc': (c_int, [_P, c_size_t, POINTER(c_void_p)]),
    'xh_host_free': (c_int, [_P, _P]),
    'xh_memcpy_h2d_async': (c_int, [_P, _P, _P, c_size_t]),
    'xh_memcpy_d2h_async': (c_int, [_P, _P, _P, c_size_t]),
    'xh_upload_file': (c_int, [_P, _P, c_char_p, ctypes.c_uint64, c_size_t, c_int]),
    'xh_download_file': (c_int, [_P, _P, c_char_p, ctypes.c_uint64, c_size_t, c_int]),
    'xh_download_files': (c_int, [_P, c_int, POINTER(c_void_p), POINTER(c_char_p), POINTER(ctypes.c_uint64), POINTER(c_size_t)]),
    'xh_memset': (c_int, [_P, _P, c_int, c_size_t]),
    'xh_sync': (c_int, [_P]),
    'xh_gather_rows': (c_int, [_P, _P, _P, c_int64, c_int64, _P]),
    'xh_scatter_rows': (c_int, [_P, _P, _P, c_int64, c_int64, _P]),
    'xh_transpose': (c_int, [_P, _P, c_int64, c_int64, _P]),
    'xh_timing_reset': (c_int, [_P]),
    'xh_timing_enable': (c_int, [_P, c_int]),
    'xh_timing_get': (c_int, [_P, c_char_p, POINTER(c_double), POINTER(c_int64)]),
    'xh_pm_pet': (c_int, [_P, POINTER(PmTables), c_int64, c_int32, c_int32, c_int32, _P, c_int32, c_int32,
                          _P, _P, _P, _P, _P, _P, _P, _P, _P, _P]),
    'xh_abcd': (c_int, [_P, c_int64, c_int32, c_int32, c_int32, _P, _P, c_int64, _P, _P, _P, _P, _P, _P, _P, _P, _P]),
    'xh_route_plan_create': (c_int, [_P, c_int64, _P, _P, _P, POINTER(c_void_p)]),
    'xh_route_plan_destroy': (None, [_P]),
    'xh_route_plan_info': (c_int, [_P, POINTER(c_int64)]),
    'xh_route_plan_stats': (c_int, [_P, c_int64, _P, POINTER(c_int64)]),
    'xh_route_plan_rsum_info': (c_int, [_P, POINTER(c_int64)]),
    'xh_route_plan_prepare': (c_int, [_P, _P, _P, _P, c_double]),
    'xh_mrtm_downstream': (c_int, [c_int64, c_int32, c_int32, _P, _P, _P, _P, _P]),
    'xh_mrtm_upstream': (c_int, [c_int64, c_int32, c_int32, _P, _P, _P, _P, _P]),
    'xh_mrtm_um_csr': (c_int, [c_int64, _P, _P, _P, _P]),
    'xh_route_series': (c_int, [_P, _P, c_int32, c_int32, _P, c_double, _P, _P, _P, _P, _P, _P, _P, _P, _P, c_int32]),
    'xh_run_fused': (c_int, [_P, POINTER(FusedArgs)]),
    'xh_calib_objective': (c_int, [_P, c_int64, c_int32, c_int32, c_int32, c_int32, _P, _P, _P, _P, _P, _P, _P, _P]),
    'xh_calib_objective_multi': (c_int, [_P, c_int32, _P, c_int32, c_int32, c_int32, c_int32, _P, _P, _P, _P, _P, _P, _P, _P]),
    'xh_calib_de_create': (c_int, [_P, c_int32, _P, _P, c_int32, c_int32, c_int32, c_int32, _P, _P, _P, _P, _P, _P, _P,
                                   c_uint64, POINTER(c_void_p)]),
    'xh_calib_de_destroy': (None, [_P]),
    'xh_calib_de_init': (c_int, [_P]),
    'xh_calib_de_step': (c_int, [_P, c_int32, c_double, c_double, c_double, c_double, c_double, POINTER(c_int32)]),
    'xh_calib_de_result': (c_int, [_P, _P, _P, _P, _P, _P]),
    'xh_calib_de_state': (c_int, [_P, c_int32, _P, _P]),
    'xh_calib_de_set_state': (c_int, [_P, _P, _P, c_int32]),
    'xh_comm_unique_id': (c_int, [ctypes.c_char_p, c_size_t]),
    'xh_comm_create': (c_int, [_P, c_int32, c_int32, ctypes.c_char_p, c_size_t, POINTER(c_void_p)]),
    'xh_comm_destroy': (None, [_P]),
    'xh_comm_info': (c_int, [_P, POINTER(c_int64)]),
    'xh_comm_gather_rows': (c_int, [_P, _P, c_int32, c_int32, _P, c_int64, _P, _P, _P]),
    'xh_comm_gather_rows_side': (c_int, [_P, _P, c_int32, c_int32, _P, c_int64, _P, _P, _P]),
    'xh_comm_join': (c_int, [_P]),
    'xh_mark_begin': (c_int, [_P, c_char_p]),
    'xh_mark_end': (c_int, [_P]),
    'xh_agg_time': (c_int, [_P, c_int64, c_int32, c_int32, c_int32, _P, _P, _P]),
    'xh_agg_spatial': (c_int, [_P, c_int64, c_int32, c_int32, _P, _P, _P]),
    'xh_nan_to_num': (c_int, [_P, _P, c_int64]),
    'xh_drought_thresholds': (c_int, [_P, c_int64, c_int32, c_int32, c_int32, c_int32, c_int32, c_int32, c_double, _P, _P]),
    'xh_drought_stats': (c_int, [_P, c_int64, c_int32, c_int32, _P, _P, _P, _P, _P]),
    'xh_synth_forcing': (c_int, [_P, c_uint64, c_double, c_int64, c_int32, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P]),
}

_lib = None


ABI_VERSION = 5        # xh_abi_version() of the library these signatures describe


def lib():
    """Load libxanthos_hip.so (once) and declare every signature. Raises HipUnavailable if it is not built."""
    global _lib
    if _lib is None:
        if not os.path.isfile(LIB_PATH):
            raise HipUnavailable(
                '{} not found: build it with `python -c "import __graft_entry__ as g; g.build()"` or '
                '`make -C xanthos_amd/csrc` (hipcc --offload-arch=gfx950). There is no CPU fallback.'.format(LIB_PATH))
        try:
            handle = ctypes.CDLL(LIB_PATH)
        except OSError as exc:
            raise HipUnavailable('cannot load {}: {}'.format(LIB_PATH, exc))
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(handle, name)      # AttributeError here = header / library mismatch
            fn.restype = res
            fn.argtypes = args
        if handle.xh_abi_version() != ABI_VERSION:
            raise HipUnavailable('{} has ABI version {}, this package binds version {}: rebuild it (make -C xanthos_amd/csrc)'
                                 .format(LIB_PATH, handle.xh_abi_version(), ABI_VERSION))
        _lib = handle
    return _lib


def _host_ptr(arr):
    return arr.ctypes.data_as(c_void_p)


def as_f64(a):
    return np.ascontiguousarray(a, dtype=np.float64)


class DeviceArray:
    """A float64 (or raw) buffer in HBM owned by a Context."""

    def __init__(self, ctx, shape, dtype=np.float64):
        self.ctx = ctx
        self.shape = tuple(int(s) for s in (shape if isinstance(shape, (tuple, list)) else (shape,)))
        self.dtype = np.dtype(dtype)
        self.nbytes = int(np.prod(self.shape, dtype=np.int64)) * self.dtype.itemsize
        p = c_void_p()
        ctx._check(lib().xh_malloc(ctx.handle, self.nbytes, byref(p)))
        self.ptr = p.value
        ctx._live.add(self)

    @property
    def size(self):
        return int(np.prod(self.shape, dtype=np.int64))

    def upload(self, host):
        host = np.ascontiguousarray(host, dtype=self.dtype)
        if host.nbytes != self.nbytes:
            raise ValueError('size mismatch: host {} vs device {}'.format(host.shape, self.shape))
        self.ctx._check(lib().xh_memcpy_h2d(self.ctx.handle, self.ptr, _host_ptr(host), self.nbytes))
        return self

    def download(self, out=None):
        if out is None:
            out = np.empty(self.shape, dtype=self.dtype)
        if out.nbytes != self.nbytes or not out.flags.c_contiguous:
            raise ValueError('bad output buffer')
        self.ctx._check(lib().xh_memcpy_d2h(self.ctx.handle, _host_ptr(out), self.ptr, self.nbytes))
        return out

    def zero(self):
        self.ctx._check(lib().xh_memset(self.ctx.handle, self.ptr, 0, self.nbytes))
        return self

    def free(self):
        if self.ptr is not None and self.ctx.handle is not None:
            lib().xh_free(self.ctx.handle, self.ptr)
            self.ctx._live.discard(self)
        self.ptr = None

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass


def _dptr(x):
    """Device pointer of a DeviceArray / raw int / None."""
    if x is None:
        return None
    if isinstance(x, DeviceArray):
        return x.ptr
    return int(x)


class Context:
    """One HIP device + stream (xh_ctx). Not thread-safe; use one per host thread."""

    def __init__(self, device=0):
        L = lib()
        h = c_void_p()
        rc = L.xh_ctx_create(int(device), byref(h))
        if rc != 0:
            msg = L.xh_last_error(None)
            raise HipUnavailable('xh_ctx_create(device={}) failed: {}. There is no CPU fallback.'.format(
                device, msg.decode() if msg else rc))
        self.handle = h.value
        self.device = int(device)
        self._live = set()
        self._pinned = {}

    # ---- plumbing
    def _check(self, rc):
        if rc != 0:
            msg = lib().xh_last_error(self.handle)
            raise HipError('libxanthos_hip error {}: {}'.format(rc, msg.decode() if msg else ''))

    def name(self):
        buf = ctypes.create_string_buffer(256)
        self._check(lib().xh_device_name(self.handle, buf, 256))
        return buf.value.decode()

    def cu_count(self):
        """Compute units of the device (xh_device_name ends in "(<arch>, <n> CUs)")."""
        import re
        m = re.search(r'(\d+) CUs\)', self.name())
        return int(m.group(1)) if m else 256

    def close(self):
        if self.handle is not None:
            for a in list(self._live):
                a.free()
            for p in list(self._pinned.values()):
                lib().xh_host_free(self.handle, p)
            self._pinned = {}
            lib().xh_ctx_destroy(self.handle)
            self.handle = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def sync(self):
        self._check(lib().xh_sync(self.handle))

    def empty(self, shape, dtype=np.float64):
        return DeviceArray(self, shape, dtype)

    def upload(self, host, dtype=np.float64):
        host = np.ascontiguousarray(host, dtype=dtype)
        return DeviceArray(self, host.shape, dtype).upload(host)

    def pinned(self, shape, dtype=np.float64):
        """A numpy array in page-locked host memory (xh_host_alloc); release it with free_pinned()."""
        shape = tuple(int(s) for s in (shape if isinstance(shape, (tuple, list)) else (shape,)))
        nbytes = int(np.prod(shape, dtype=np.int64)) * np.dtype(dtype).itemsize
        p = c_void_p()
        self._check(lib().xh_host_alloc(self.handle, nbytes, byref(p)))
        buf = (ctypes.c_char * max(nbytes, 1)).from_address(p.value)
        arr = np.frombuffer(buf, dtype=dtype, count=int(np.prod(shape, dtype=np.int64))).reshape(shape)
        self._pinned[arr.ctypes.data] = p.value
        return arr

    def free_pinned(self, arr):
        p = self._pinned.pop(arr.ctypes.data, None)
        if p is not None and self.handle is not None:
            self._check(lib().xh_host_free(self.handle, p))

    def h2d_async(self, dst, host):
        """Enqueue host -> device without waiting; ``host`` must stay alive and unchanged until sync()."""
        self._check(lib().xh_memcpy_h2d_async(self.handle, _dptr(dst), _host_ptr(host), host.nbytes))

    def d2h_async(self, host, src):
        self._check(lib().xh_memcpy_d2h_async(self.handle, _host_ptr(host), _dptr(src), host.nbytes))

    def upload_file(self, dst, path, offset, nbytes, threads=0):
        """``nbytes`` of file ``path`` from byte ``offset`` -> DeviceArray ``dst`` (xh_upload_file: the range is mapped
        read-only and copied out of the mapping; no host copy of the data)."""
        if nbytes != dst.nbytes:
            raise ValueError('size mismatch: file range {} vs device {}'.format(nbytes, dst.nbytes))
        self._check(lib().xh_upload_file(self.handle, _dptr(dst), os.fsencode(path), int(offset), int(nbytes), threads))
        return dst

    def download_file(self, src, path, offset, threads=0):
        """DeviceArray ``src`` -> bytes ``offset ...`` of file ``path`` (xh_download_file)."""
        self._check(lib().xh_download_file(self.handle, _dptr(src), os.fsencode(path), int(offset), src.nbytes, threads))

    def save_npy(self, path, src):
        """np.save(path, src.download()) without the host array: header here, body by xh_download_file."""
        self.save_npy_many([(path, src)])

    def save_npy_many(self, items):
        """[(path, DeviceArray), ...] -> .npy files, the bodies written side by side (xh_download_files, <= 16 per call)."""
        items = list(items)
        for k in range(0, len(items), 16):
            part = items[k:k + 16]
            offsets = []
            for path, src in part:
                with open(path, 'wb') as fh:
                    np.lib.format.write_array_header_1_0(fh, {'descr': np.lib.format.dtype_to_descr(np.dtype(src.dtype)),
                                                              'fortran_order': False, 'shape': tuple(src.shape)})
                    offsets.append(fh.tell())
            n = len(part)
            srcs = (c_void_p * n)(*[_dptr(src) for _, src in part])
            paths = (c_char_p * n)(*[os.fsencode(path) for path, _ in part])
            offs = (ctypes.c_uint64 * n)(*offsets)
            sizes = (c_size_t * n)(*[src.nbytes for _, src in part])
            self._check(lib().xh_download_files(self.handle, n, srcs, paths, offs, sizes))

    def timing_reset(self):
        self._check(lib().xh_timing_reset(self.handle))

    def comm_join(self):
        """Order the context's stream behind the side gather (xh_comm_join)."""
        self._check(lib().xh_comm_join(self.handle))

    def mark_begin(self, name):
        self._check(lib().xh_mark_begin(self.handle, name.encode()))

    def mark_end(self):
        self._check(lib().xh_mark_end(self.handle))

    def timing(self, name):
        ms, n = c_double(), c_int64()
        self._check(lib().xh_timing_get(self.handle, name.encode(), byref(ms), byref(n)))
        return ms.value, n.value

    def gather_rows(self, src, rows_dev, nrows, ncols, dst):
        self._check(lib().xh_gather_rows(self.handle, _dptr(src), _dptr(rows_dev), nrows, ncols, _dptr(dst)))

    def scatter_rows(self, src, rows_dev, nrows, ncols, dst):
        self._check(lib().xh_scatter_rows(self.handle, _dptr(src), _dptr(rows_dev), nrows, ncols, _dptr(dst)))

    def transpose(self, src, rows, cols, dst):
        self._check(lib().xh_transpose(self.handle, _dptr(src), rows, cols, _dptr(dst)))

    # ---- Penman-Monteith
    @staticmethod
    def _pm_tables(tables):
        """dict of host arrays (cL ... laimax) -> (PmTables, arrays to keep alive)."""
        keep = []
        t = PmTables()
        t.nlcs = int(np.asarray(tables['cL']).shape[0])
        for name, _ in PmTables._fields_[1:]:
            a = as_f64(tables[name]).ravel()
            want = t.nlcs * (12 if name in ('alpha', 'lai', 'laimin', 'laimax') else 1)
            if a.size != want:
                raise ValueError('PM table {} has {} values, expected {}'.format(name, a.size, want))
            keep.append(a)
            setattr(t, name, a.ctypes.data_as(POINTER(c_double)))
        return t, keep

    def run_fused(self, *, tables, ncell, nmonths, start_year, lc_years, water_idx, snow_idx, tas, tmin, rhs, wind, rsds,
                  rlds, tairprev, lct, elev, abcd_spinup, n_groups, basin_index, par_index, npar_rows, pars, precip,
                  abcd_tmin, pet, aet, q, sav, plan=None, routing_spinup=0, ndays=None, dt=10800.0, flow_dist=None,
                  velocity=None, area=None, S0=None, chs=None, avg=None, route_flags=0, block_months=0, mode=0):
        """PM -> ABCD -> MRTM as one pipelined call (xh_run_fused); arguments as in pm_pet / abcd / route_series.
        mode 1: the routing kernel starts after the first max(spin-ups) months and is fed the rest while it runs."""
        t, keep = self._pm_tables(tables)
        lcy = np.ascontiguousarray(lc_years, dtype=np.int32)
        bi = np.ascontiguousarray(basin_index, dtype=np.int32)
        pi = np.ascontiguousarray(par_index, dtype=np.int32)
        nd = None if ndays is None else np.ascontiguousarray(ndays, dtype=np.int32)
        if plan is not None and (nd is None or nd.size != nmonths):
            raise ValueError('ndays must have nmonths entries')
        a = FusedArgs()
        a.ncell, a.nmonths, a.start_year = ncell, nmonths, start_year
        a.pm = ctypes.pointer(t)
        a.n_lc_years, a.h_lc_years, a.water_idx, a.snow_idx = len(lcy), lcy.ctypes.data, water_idx, snow_idx
        for name, v in (('d_tas', tas), ('d_tmin', tmin), ('d_rhs', rhs), ('d_wind', wind), ('d_rsds', rsds),
                        ('d_rlds', rlds), ('d_tairprev', tairprev), ('d_lct', lct), ('d_elev', elev), ('d_pars', pars),
                        ('d_precip', precip), ('d_abcd_tmin', abcd_tmin), ('d_flow_dist', flow_dist),
                        ('d_velocity', velocity), ('d_area', area), ('d_S0', S0), ('d_pet', pet), ('d_aet', aet),
                        ('d_q', q), ('d_sav', sav), ('d_chstorage', chs), ('d_avgchflow', avg)):
            setattr(a, name, _dptr(v))
        a.abcd_spinup, a.n_groups, a.npar_rows = abcd_spinup, n_groups, npar_rows
        a.h_basin_index, a.h_par_index = bi.ctypes.data, pi.ctypes.data
        a.plan = None if plan is None else plan.handle
        a.routing_spinup, a.dt, a.route_flags, a.block_months = routing_spinup, float(dt), int(route_flags), int(block_months)
        a.h_ndays = None if nd is None else nd.ctypes.data
        a.mode = int(mode)
        self._check(lib().xh_run_fused(self.handle, byref(a)))

    def pm_pet(self, tables, ncell, nmonths, start_year, lc_years, water_idx, snow_idx, tas, tmin, rhs, wind, rsds,
               rlds, tairprev, lct, elev, pet):
        """tables: dict of host arrays (cL ... laimax); the rest device arrays / pointers."""
        t, keep = self._pm_tables(tables)
        lcy = np.ascontiguousarray(lc_years, dtype=np.int32)
        self._check(lib().xh_pm_pet(self.handle, byref(t), ncell, nmonths, start_year, len(lcy), _host_ptr(lcy),
                                    water_idx, snow_idx, _dptr(tas), _dptr(tmin), _dptr(rhs), _dptr(wind),
                                    _dptr(rsds), _dptr(rlds), _dptr(tairprev), _dptr(lct), _dptr(elev), _dptr(pet)))

    # ---- ABCD
    def abcd(self, ncell, nmonths, spinup, n_groups, basin_index, par_index, npar_rows, pars, pet, precip, tmin,
             aet, q, sav, sm0=None, gw0=None):
        bi = np.ascontiguousarray(basin_index, dtype=np.int32)
        pi = np.ascontiguousarray(par_index, dtype=np.int32)
        self._check(lib().xh_abcd(self.handle, ncell, nmonths, spinup, n_groups, _host_ptr(bi), _host_ptr(pi),
                                  npar_rows, _dptr(pars), _dptr(pet), _dptr(precip), _dptr(tmin), _dptr(aet),
                                  _dptr(q), _dptr(sav), _dptr(sm0), _dptr(gw0)))

    # ---- MRTM
    def route_plan(self, indptr, indices, sign):
        return RoutePlan(self, indptr, indices, sign)

    def route_series(self, plan, nmonths, spinup_months, ndays, dt, flow_dist, velocity, area, runoff, S0, chs, avg,
                     S_end=None, F_end=None, flags=XH_ROUTE_DEFAULT):
        nd = np.ascontiguousarray(ndays, dtype=np.int32)
        if nd.size != nmonths:
            raise ValueError('ndays must have nmonths entries')
        self._check(lib().xh_route_series(self.handle, plan.handle, nmonths, spinup_months, _host_ptr(nd), float(dt),
                                          _dptr(flow_dist), _dptr(velocity), _dptr(area), _dptr(runoff), _dptr(S0),
                                          _dptr(chs), _dptr(avg), _dptr(S_end), _dptr(F_end), int(flags)))

    # ---- calibration objective
    def calib_objective(self, ncell_b, nmonths, spinup, pars, pet_t, precip_t, tmin_t, area, obs, want_series=False):
        pars = as_f64(pars)
        nmem, npar = pars.shape
        obs = as_f64(obs)
        if obs.size != nmonths:
            raise ValueError('obs must have nmonths entries')
        ed = np.empty(nmem)
        series = np.empty((nmem, nmonths)) if want_series else None
        self._check(lib().xh_calib_objective(self.handle, ncell_b, nmonths, spinup, nmem, npar, _host_ptr(pars),
                                             _dptr(pet_t), _dptr(precip_t), _dptr(tmin_t), _dptr(area),
                                             _host_ptr(obs), _host_ptr(ed),
                                             _host_ptr(series) if want_series else None))
        return (ed, series) if want_series else ed

    def calib_objective_multi(self, ncells, nmonths, spinup, pars, pet_t, precip_t, tmin_t, area, obs, want_series=False):
        """Several basins at once. pars [nb, nmem, npar]; pet_t / precip_t / tmin_t / area: lists of DeviceArrays
        (tmin_t / area may be None); obs [nb, nmonths]. Returns ed [nb, nmem] (and series [nb, nmem, nmonths])."""
        pars = as_f64(pars)
        nb, nmem, npar = pars.shape
        obs = as_f64(obs)
        if obs.shape != (nb, nmonths):
            raise ValueError('obs must be [nbasins, nmonths]')
        nc = np.ascontiguousarray(ncells, dtype=np.int64)
        ptrs = lambda lst: None if lst is None else (c_void_p * nb)(*[_dptr(x) for x in lst])
        p_pet, p_pr, p_tn, p_ar = ptrs(pet_t), ptrs(precip_t), ptrs(tmin_t), ptrs(area)
        ed = np.empty((nb, nmem))
        series = np.empty((nb, nmem, nmonths)) if want_series else None
        self._check(lib().xh_calib_objective_multi(self.handle, nb, _host_ptr(nc), nmonths, spinup, nmem, npar,
                                                   _host_ptr(pars), p_pet, p_pr, p_tn, p_ar, _host_ptr(obs), _host_ptr(ed),
                                                   _host_ptr(series) if want_series else None))
        return (ed, series) if want_series else ed

    # ---- output aggregation
    def agg_time(self, ncell, ncols, group, mode, scale, src, dst):
        self._check(lib().xh_agg_time(self.handle, ncell, ncols, group, mode, _dptr(scale), _dptr(src), _dptr(dst)))

    def agg_spatial(self, ncell, ncols, n_groups, group_index, src, dst):
        gi = np.ascontiguousarray(group_index, dtype=np.int32)
        self._check(lib().xh_agg_spatial(self.handle, ncell, ncols, n_groups, _host_ptr(gi), _dptr(src), _dptr(dst)))

    def nan_to_num(self, arr):
        """In-place np.nan_to_num of a DeviceArray."""
        self._check(lib().xh_nan_to_num(self.handle, _dptr(arr), arr.size))
        return arr

    def drought_thresholds(self, ncell, nmonths, month0, nyear, nper, k_prev, k_next, gamma, hydro, thresh):
        self._check(lib().xh_drought_thresholds(self.handle, ncell, nmonths, month0, nyear, nper, k_prev, k_next,
                                                float(gamma), _dptr(hydro), _dptr(thresh)))

    def drought_stats(self, ncell, nmonths, nthresh, hydro, thresh, severity, intensity, duration):
        self._check(lib().xh_drought_stats(self.handle, ncell, nmonths, nthresh, _dptr(hydro), _dptr(thresh),
                                           _dptr(severity), _dptr(intensity), _dptr(duration)))

    # ---- bench support
    def synth_forcing(self, seed, ncell, nmonths, lat, out, nan_frac=0.001, cell_ids=None):
        """out: dict name -> DeviceArray for synth.FORCING_NAMES (or only 'tas'). cell_ids: device int64 [ncell]
        global cell of each row (None = 0..ncell-1; -1 = a row of zeros)."""
        g = lambda k: _dptr(out.get(k))
        self._check(lib().xh_synth_forcing(self.handle, int(seed), float(nan_frac), ncell, nmonths, _dptr(lat),
                                           _dptr(cell_ids), g('tas'), g('tmin'), g('rhs'), g('wind'), g('rsds'),
                                           g('rlds'), g('precip'), g('abcd_tmin')))


class RoutePlan:
    """Device-side routing layout for one UM matrix (xh_route_plan)."""

    def __init__(self, ctx, indptr, indices, sign):
        self.ctx = ctx
        ip = np.ascontiguousarray(indptr, dtype=np.int64)
        ix = np.ascontiguousarray(indices, dtype=np.int32)
        sg = np.ascontiguousarray(sign, dtype=np.int8)
        self.ncell = ip.size - 1
        h = c_void_p()
        ctx._check(lib().xh_route_plan_create(ctx.handle, self.ncell, _host_ptr(ip), _host_ptr(ix), _host_ptr(sg),
                                              byref(h)))
        self.handle = h.value

    def info(self):
        arr = (c_int64 * 16)()
        self.ctx._check(lib().xh_route_plan_info(self.handle, arr))
        keys = ('networks', 'largest_network', 'units', 'fallback_cells', 'largest_unit', 'slots', 'single_downstream',
                'flow_units', 'flow_edges', 'flow_depth', 'flow_cells', 'flow_max_imports', 'skew_max_lag',
                'last_tree_kernel', 'reroutes', 'validated')
        return dict(zip(keys, list(arr)))

    def prepare(self, flow_dist, velocity, dt):
        """xh_route_plan_prepare: host copies of flow distance and velocity [ncell] and dt.  Makes the PREPARED plan of the
        default (reassociated) routing form -- folded leaves, single running sums: both rest on which cells can fire -- or, for
        the bit-exact form, the selective plain tables when this box has learnt the grid's firing cells before.  May be
        called again: the same data is a cheap no-op, other data replaces the prepared plan."""
        L = np.ascontiguousarray(flow_dist, dtype=np.float64)
        v = np.ascontiguousarray(velocity, dtype=np.float64)
        if L.size != self.ncell or v.size != self.ncell:
            raise ValueError('flow_dist / velocity must have one value per cell')
        self.ctx._check(lib().xh_route_plan_prepare(self.ctx.handle, self.handle, _host_ptr(L), _host_ptr(v), float(dt)))

    def rsum_info(self):
        """The reassociated plan the last call ran on (xh_route_plan_rsum_info)."""
        arr = (c_int64 * 8)()
        self.ctx._check(lib().xh_route_plan_rsum_info(self.handle, arr))
        return dict(zip(('units', 'folded', 'fold_disabled', 'prepared_folded', 'pair_cells', 'prepared_pair_cells', 'guard_trips',
                         'pair_units'), list(arr)))

    def stats(self):
        """[units, 6] uint64 per-unit accounting of the last launch (needs XH_FLOW_STATS=1), or None."""
        n = c_int64(0)
        self.ctx._check(lib().xh_route_plan_stats(self.handle, 0, None, byref(n)))
        if n.value == 0:
            return None
        out = np.empty(n.value, dtype=np.uint64)
        self.ctx._check(lib().xh_route_plan_stats(self.handle, n.value, _host_ptr(out), byref(n)))
        return out.reshape(-1, 6)

    def close(self):
        if self.handle is not None and self.ctx.handle is not None:
            lib().xh_route_plan_destroy(self.handle)
        self.handle = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def comm_unique_id():
    """128-byte RCCL id made on rank 0; the launcher carries it to the other ranks (xh_comm_unique_id)."""
    buf = ctypes.create_string_buffer(128)
    _check_host(lib().xh_comm_unique_id(buf, 128))
    return buf.raw


class Comm:
    """RCCL communicator of one rank for the write-out gather (xh_comm)."""

    def __init__(self, ctx, nranks, rank, unique_id):
        self.ctx, self.nranks, self.rank = ctx, int(nranks), int(rank)
        h = c_void_p()
        # RCCL prints a version banner on stdout when a communicator is created; stdout belongs to the caller (bench.py
        # prints ONE JSON line there), so the banner is sent to stderr
        import sys
        sys.stdout.flush()
        saved = os.dup(1)
        try:
            os.dup2(2, 1)
            rc = lib().xh_comm_create(ctx.handle, self.nranks, self.rank, unique_id, len(unique_id), byref(h))
            ctypes.CDLL(None).fflush(None)          # the banner sits in C stdio's buffer until flushed
        finally:
            os.dup2(saved, 1)
            os.close(saved)
        ctx._check(rc)
        self.handle = h.value

    def info(self):
        """{'ranks', 'rank', 'sends', 'recvs'}: ncclSend / ncclRecv calls issued through this communicator (xh_comm_info)."""
        arr = (c_int64 * 4)()
        self.ctx._check(lib().xh_comm_info(self.handle, arr))
        return dict(zip(('ranks', 'rank', 'sends', 'recvs'), list(arr)))

    def gather_rows(self, local, counts, ncols, perm=None, out=None, root=0, side=False):
        """local: list of DeviceArrays [counts[rank], ncols]; on the root ``perm`` (device int64, rank-major destination
        rows) and ``out`` (list of DeviceArrays [sum(counts), ncols]). Asynchronous.  ``side``: on the context's gather
        stream, beside what the context's own stream does next (xh_comm_gather_rows_side; Context.comm_join orders the
        context's stream behind it)."""
        nvar = len(local)
        cn = np.ascontiguousarray(counts, dtype=np.int64)
        p_local = (c_void_p * nvar)(*[_dptr(x) for x in local])
        p_out = None if out is None else (c_void_p * nvar)(*[_dptr(x) for x in out])
        fn = lib().xh_comm_gather_rows_side if side else lib().xh_comm_gather_rows
        self.ctx._check(fn(self.ctx.handle, self.handle, int(root), nvar, p_local, int(ncols), _host_ptr(cn), _dptr(perm),
                           p_out))

    def close(self):
        if self.handle is not None:
            lib().xh_comm_destroy(self.handle)
        self.handle = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class CalibDE:
    """Device-side differential evolution over several basins in lock-step (xh_calib_de).

    ncells [nb]; pet_t / precip_t / tmin_t / area: lists of DeviceArrays ([nmonths, ncell_b]; tmin_t / area may be
    None) that must stay alive as long as the session; obs [nb, nmonths]; bounds [(lo, hi)] * npar; keys [nb] RNG
    stream of each basin (e.g. the basin number) so that a basin's search does not depend on its companions."""

    def __init__(self, ctx, ncells, nmonths, spinup, nmembers, bounds, pet_t, precip_t, tmin_t, area, obs, seed=0,
                 keys=None):
        self.ctx = ctx
        nc = np.ascontiguousarray(ncells, dtype=np.int64)
        self.nb, self.n, self.d = int(nc.size), int(nmembers), len(bounds)
        obs = as_f64(obs)
        if obs.shape != (self.nb, nmonths):
            raise ValueError('obs must be [nbasins, nmonths]')
        lo, hi = as_f64([b[0] for b in bounds]), as_f64([b[1] for b in bounds])
        kk = None if keys is None else np.ascontiguousarray(keys, dtype=np.uint64)
        ptrs = lambda lst: None if lst is None else (c_void_p * self.nb)(*[_dptr(x) for x in lst])
        self._keep = (pet_t, precip_t, tmin_t, area)
        h = c_void_p()
        ctx._check(lib().xh_calib_de_create(ctx.handle, self.nb, _host_ptr(nc), None if kk is None else _host_ptr(kk),
                                            nmonths, spinup, self.n, self.d, ptrs(pet_t), ptrs(precip_t), ptrs(tmin_t),
                                            ptrs(area), _host_ptr(obs), _host_ptr(lo), _host_ptr(hi),
                                            int(seed) & 0xFFFFFFFFFFFFFFFF, byref(h)))
        self.handle = h.value

    def init(self):
        self.ctx._check(lib().xh_calib_de_init(self.handle))

    def step(self, ngen=1, tol=0.01, atol=0.0, mutation=(0.5, 1.0), recombination=0.7):
        """Run ``ngen`` generations; returns the number of basins still searching."""
        left = c_int32(0)
        self.ctx._check(lib().xh_calib_de_step(self.handle, int(ngen), float(tol), float(atol), float(mutation[0]),
                                               float(mutation[1]), float(recombination), byref(left)))
        return left.value

    def result(self):
        """(x [nb, d], fun [nb], nfev [nb], nit [nb], active [nb])."""
        x, fun = np.empty((self.nb, self.d)), np.empty(self.nb)
        nfev, nit, act = np.empty(self.nb, dtype=np.int64), np.empty(self.nb, dtype=np.int32), \
            np.empty(self.nb, dtype=np.int32)
        self.ctx._check(lib().xh_calib_de_result(self.handle, _host_ptr(x), _host_ptr(fun), _host_ptr(nfev),
                                                 _host_ptr(nit), _host_ptr(act)))
        return x, fun, nfev, nit, act

    def state(self, which=0):
        """which: 0 population + energies, 1 last trial (unit cube) + energies, 2 last trial scaled + energies."""
        v, e = np.empty((self.nb, self.n, self.d)), np.empty((self.nb, self.n))
        self.ctx._check(lib().xh_calib_de_state(self.handle, int(which), _host_ptr(v), _host_ptr(e)))
        return v, e

    def set_state(self, pop, energies, generation=0):
        pop, energies = as_f64(pop), as_f64(energies)
        if pop.shape != (self.nb, self.n, self.d) or energies.shape != (self.nb, self.n):
            raise ValueError('bad state shape')
        self.ctx._check(lib().xh_calib_de_set_state(self.handle, _host_ptr(pop), _host_ptr(energies), int(generation)))

    def close(self):
        if self.handle is not None and self.ctx.handle is not None:
            lib().xh_calib_de_destroy(self.handle)
        self.handle = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


# ---- host-only topology helpers (no device needed)
def _topo_args(coords):
    coords = np.asarray(coords)
    ids = np.ascontiguousarray(coords[:, 0], dtype=np.int64)
    ilon = np.ascontiguousarray(coords[:, 3], dtype=np.int32)
    ilat = np.ascontiguousarray(coords[:, 4], dtype=np.int32)
    return ids, ilon, ilat


def _check_host(rc):
    if rc != 0:
        msg = lib().xh_last_error(None)
        raise HipError('libxanthos_hip error {}: {}'.format(rc, msg.decode() if msg else ''))


def mrtm_downstream(coords, flowdir, nrow, ncol):
    ids, ilon, ilat = _topo_args(coords)
    fd = as_f64(flowdir)
    out = np.empty(ids.size, dtype=np.int64)
    _check_host(lib().xh_mrtm_downstream(ids.size, nrow, ncol, _host_ptr(ids), _host_ptr(ilon), _host_ptr(ilat),
                                         _host_ptr(fd), _host_ptr(out)))
    return out


def mrtm_upstream(coords, dsid, nrow, ncol):
    ids, ilon, ilat = _topo_args(coords)
    ds = np.ascontiguousarray(dsid, dtype=np.int64)
    out = np.empty((ids.size, 9), dtype=np.int64)
    _check_host(lib().xh_mrtm_upstream(ids.size, nrow, ncol, _host_ptr(ids), _host_ptr(ilon), _host_ptr(ilat),
                                       _host_ptr(ds), _host_ptr(out)))
    return out


def mrtm_um_csr(upid):
    up = np.ascontiguousarray(upid, dtype=np.int64)
    n = up.shape[0]
    nnz = n + int(up[:, 8].sum())
    indptr = np.empty(n + 1, dtype=np.int64)
    indices = np.empty(nnz, dtype=np.int32)
    sign = np.empty(nnz, dtype=np.int8)
    _check_host(lib().xh_mrtm_um_csr(n, _host_ptr(up), _host_ptr(indptr), _host_ptr(indices), _host_ptr(sign)))
    return indptr, indices, sign


_contexts = {}


def get_context(device=0):
    """Process-wide context per device."""
    ctx = _contexts.get(device)
    if ctx is None or ctx.handle is None:
        ctx = _contexts[device] = Context(device)
    return ctx


def device_count():
    n = c_int(0)
    lib().xh_device_count(byref(n))
    return n.value
