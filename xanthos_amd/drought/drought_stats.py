"""Drought statistics (mirror of xanthos/drought/drought_stats.py:22-171) on the GPU.

Same class and methods as the reference -- ``DroughtStats(settings, runoff, soil_moisture)``, ``calculate_thresholds``,
``droughtstats``, ``getthresh`` -- with the reference's array shapes at the interface (``[ntime, ngrid]`` hydrology,
``[K, ngrid]`` thresholds).  The two array computations run as HIP kernels (csrc/xh_drought.hip) on ``[ncell, nmonths]``
arrays, the layout the pipeline already holds in HBM; ``DroughtStats`` itself is handed ``[ncell, nmonths]`` arrays
(host or device) by ``Components.drought`` and never transposes on the host.
"""
import logging
import math
import os

import numpy as np

from .. import _hip


def quantile_plan(nsample, q):
    """(k_prev, k_next, gamma) of numpy's ``linear`` quantile for ``nsample`` values.

    Follows numpy/lib/_function_base_impl.py (numpy 1.22 - 2.x): virtual index ``(n - 1) * q``; its floor and the next
    index bracket it, both clamped to the last element when the virtual index reaches it; the weight is the
    fractional part.  Python floats are IEEE doubles, so these are the numbers numpy computes.
    """
    vi = (nsample - 1) * q
    if vi >= nsample - 1:
        return nsample - 1, nsample - 1, 0.0 if not math.isnan(vi) else float('nan')
    if vi < 0:
        return 0, 0, vi - 0.0          # numpy keeps the (negative) weight with both indices at 0: a + 0 * g = a
    prev = math.floor(vi)
    return int(prev), int(prev) + 1, vi - prev


def _device_rows(ctx, arr):
    """[ncell, nmonths] DeviceArray from a host array or a DeviceArray (no copy in the second case)."""
    if isinstance(arr, _hip.DeviceArray):
        return arr, False
    return ctx.upload(np.ascontiguousarray(arr, dtype=np.float64)), True


def thresholds_rows(ctx, hydro_rows, month0, nmonths_ref, nper, quantile=0.1):
    """getthresh on a ``[ncell, nmonths]`` array (host or device): thresholds ``[nper, ncell]`` (host)."""
    src, mine = _device_rows(ctx, hydro_rows)
    ncell, nmonths = src.shape
    nyear = int(nmonths_ref / nper)                                  # drought_stats.py:166
    if nyear * nper != nmonths_ref:
        raise ValueError('cannot reshape array of size {} into shape ({},{},{})'.format(nmonths_ref * ncell, nyear, nper, ncell))
    q = (quantile * 100) / 100.0                                      # np.percentile(x, quantile * 100): q / 100 again
    k_prev, k_next, gamma = quantile_plan(nyear, q)
    out = ctx.empty((nper, ncell))
    ctx.drought_thresholds(ncell, nmonths, month0, nyear, nper, k_prev, k_next, gamma, src, out)
    host = out.download()
    out.free()
    if mine:
        src.free()
    return host


def droughtstats_rows(ctx, hydro_rows, threshvals, keep_on_device=False):
    """droughtstats on a ``[ncell, nmonths]`` array: (S, I, D), each ``[ncell, nmonths]``."""
    src, mine = _device_rows(ctx, hydro_rows)
    ncell, nmonths = src.shape
    th = np.ascontiguousarray(threshvals, dtype=np.float64)
    if th.ndim != 2 or th.shape[1] != ncell:
        raise ValueError('thresholds have shape {}, expected [K, {}]'.format(th.shape, ncell))
    d_th = ctx.upload(th)
    outs = [ctx.empty((ncell, nmonths)) for _ in range(3)]
    ctx.drought_stats(ncell, nmonths, th.shape[0], src, d_th, *outs)
    if keep_on_device:
        res = tuple(outs)
    else:
        res = tuple(o.download() for o in outs)
        for o in outs:
            o.free()
    d_th.free()
    if mine:
        src.free()
    return res


class DroughtStats:
    """Drought metrics after Sheffield and Wood (2008); see the reference class for the definitions."""
    MONTHS_IN_YEAR = 12

    def __init__(self, settings, runoff, soil_moisture):
        """``runoff`` / ``soil_moisture``: ``[ncell, nmonths]`` (host arrays or DeviceArrays)."""
        from ..data_writer.out_writer import OutWriter
        var = settings.drought_var.lower()
        if var == 'q':
            rows = runoff
        elif var == 'soilmoisture':
            rows = soil_moisture
        else:
            raise ValueError("Invalid drought variable specified (must be 'q' or 'soilmoisture')")
        self.ctx = _hip.get_context(getattr(settings, 'device', 0))
        output_path = os.path.join(settings.OutputFolder, 'drought_{}_{}'.format('{}', settings.OutputNameStr))
        out_writer = OutWriter(settings, 0, {})
        os.makedirs(settings.OutputFolder, exist_ok=True)
        if settings.drought_thresholds is None:
            logging.info('\tCalculating drought thresholds')
            self.thresholds = self._thresholds_rows(rows, settings)
            np.save(output_path.format('thresholds'), self.thresholds)
        else:
            logging.info('\tCalculating drought statistics')
            threshvals = np.load(settings.drought_thresholds)
            self.severity, self.intensity, self.duration = droughtstats_rows(self.ctx, rows, threshvals)
            nm = self.severity.shape[1]
            cols = [str(x) for x in range(nm)]                       # the reference names the columns 0 .. ntime-1
            for varname, arr in zip(('severity', 'intensity', 'duration'),
                                    (self.severity, self.intensity, self.duration)):
                # a fresh DataFrame's index starts at 0 (drought_stats.py:64-65): ids 0 .. ncell-1, unlike write()'s 1-based ids
                out_writer.write_data(output_path.format(varname), varname, arr, cols, first_id=0)

    @classmethod
    def _window(cls, settings):
        """Month slice of the reference period exactly as drought_stats.py:77-82 forms it."""
        syear, eyear = settings.threshold_start_year, settings.threshold_end_year
        smonth = (syear - settings.StartYear) * cls.MONTHS_IN_YEAR
        emonth = (eyear + 1 - syear) * cls.MONTHS_IN_YEAR            # sic: a length used as an end index (:80)
        return smonth, emonth

    def _thresholds_rows(self, rows, settings):
        smonth, emonth = self._window(settings)
        nmonths = rows.shape[1]
        stop = min(emonth, nmonths)
        return thresholds_rows(self.ctx, rows, smonth, max(stop - smonth, 0), settings.threshold_nper)

    @classmethod
    def calculate_thresholds(cls, histout, settings):
        """``histout`` [ntime x ngrid] -> thresholds [nper x ngrid] (drought_stats.py:69-83)."""
        smonth, emonth = cls._window(settings)
        return cls.getthresh(np.asarray(histout)[smonth:emonth, :], settings.threshold_nper)

    def droughtstats(self, hydroout, threshvals):
        """(S, I, D), each [ntime x ngrid], from ``hydroout`` [ntime x ngrid] and ``threshvals`` [K x ngrid]."""
        ctx = getattr(self, 'ctx', None) or _hip.get_context(0)
        d_t = ctx.upload(np.ascontiguousarray(hydroout, dtype=np.float64))          # [ntime, ngrid]
        ntime, ngrid = d_t.shape
        d_rows = ctx.empty((ngrid, ntime))
        ctx.transpose(d_t, ntime, ngrid, d_rows)
        d_t.free()
        outs = droughtstats_rows(ctx, d_rows, threshvals, keep_on_device=True)
        d_rows.free()
        res = []
        for o in outs:
            d_tt = ctx.empty((ntime, ngrid))
            ctx.transpose(o, ngrid, ntime, d_tt)
            res.append(d_tt.download())
            d_tt.free()
            o.free()
        return tuple(res)

    @staticmethod
    def getthresh(histout, nper, quantile=0.1):
        """Quantile thresholds [nper x ngrid] from ``histout`` [ntime x ngrid] (drought_stats.py:150-171)."""
        ctx = _hip.get_context(0)
        h = np.ascontiguousarray(histout, dtype=np.float64)
        ntime, ngrid = h.shape
        d_t = ctx.upload(h)
        d_rows = ctx.empty((ngrid, ntime))
        ctx.transpose(d_t, ntime, ngrid, d_rows)
        d_t.free()
        out = thresholds_rows(ctx, d_rows, 0, ntime, nper, quantile)
        d_rows.free()
        return out
