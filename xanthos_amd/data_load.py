"""Input loading for the pm / abcd / mrtm path (host side, runs once).

Mirrors the parts of xanthos/data_reader/data_load.py that feed the hot path and keeps every load-time transform that
changes its inputs:

* area x 0.01 (ha -> km2, :48); coordinates table [id, lon, lat, ilon, ilat] (:51); basin ids (:54)
* PM forcings pass through nan_to_num (:120-125) -- on the device after upload (``xh_nan_to_num``), on the host only if
  ``settings.device_transforms = False``; ``tairprev`` is tas shifted by one CELL, zeros for cell 0 (:127-129);
  land cover and elevation nan_to_num (:132,:135); 13 per-class parameters + 4 per-(class, month) tables (:94-117)
* precipitation keeps NaN (:186); ABCD tmin nan_to_num (:194-195), optional
* routing: flow distance < 1000 -> 1000 (:204-205), velocity < 0 -> 0 (:207-208), 2-D DRT maps flattened with the
  reference's ``vectorize`` (:415-425), zero initial channel storage in historic mode (:427-438)

Any setting may be an in-memory ndarray instead of a path (data_load.py:305-307), which is how tests and benchmarks
inject synthetic forcing.
"""
import os

import numpy as np

from .ini_reader import ValidationException


def load_file(fn, header_num=0, key=None, mmap=False):
    """.npy / .csv / .txt / .nc (NetCDF classic) / .mat reader, same dispatch as data_load.py:342-390.
    mmap: a .npy comes back as a read-only memory map (np.load(mmap_mode='r')): nothing is read until someone looks, and
    the pipeline sends the file's bytes to the GPU itself (xh_upload_file)."""
    if isinstance(fn, np.ndarray):
        return fn
    if not os.path.isfile(fn):
        raise IOError('Error: File does not exist:', fn)
    if fn.endswith('.npy'):
        return np.load(fn, mmap_mode='r' if mmap else None)
    if fn.endswith('.mat'):
        import scipy.io as sio
        return sio.loadmat(fn)[key]
    if fn.endswith('.nc'):
        import scipy.io as sio
        grp = sio.netcdf_file(fn, 'r', mmap=False)
        data = grp.variables[key][:].copy()
        grp.close()
        if data.dtype.byteorder == '>':            # NetCDF classic is big-endian (data_load.py:381-384)
            data = data.byteswap().view(data.dtype.newbyteorder())
        return data
    if fn.endswith('.csv') or fn.endswith('.txt'):
        delim = ',' if fn.endswith('.csv') else ' '
        try:            # numpy's C tokenizer: ~15x faster than genfromtxt on the 67,420-row grid tables; same values
            return np.loadtxt(fn, delimiter=delim, skiprows=header_num, dtype=float)
        except ValueError:      # missing fields (the reference fills them with 0), ragged or non-numeric columns
            return np.genfromtxt(fn, delimiter=delim, skip_header=header_num, filling_values='0')
    raise RuntimeError('File {} has unrecognized extension'.format(fn))


def _present(f):
    return isinstance(f, np.ndarray) or (isinstance(f, str) and os.path.isfile(f))


def vectorize(data, ngridrow, ngridcol, map_index, skip=68):
    """2-D DRT map (rows north to south) -> per-cell vector (data_load.py:415-425)."""
    # row i of the map (south to north after the flip) lands on grid row i + skip; map_index addresses the grid in
    # column-major order, so cell k reads new[r, c] with r = map_index % ngridrow, c = map_index // ngridrow: one
    # index gather per cell instead of assembling (and re-ordering) the whole 360 x 720 grid
    data = np.asarray(data, dtype=float)
    map_index = np.asarray(map_index)
    r, c = map_index % ngridrow, map_index // ngridrow
    src = data.shape[0] - 1 - (r - skip)
    inside = (r >= skip) & (r < skip + data.shape[0])
    out = np.full(map_index.shape, -9999.0)
    out[inside] = data[src[inside], c[inside]]
    return out


class DataLoader:
    """Arrays the three plugins need, as attributes with the reference's names."""

    def __init__(self, s):
        self.s = s
        self.area = np.asarray(load_file(s.Area), dtype=float).reshape(-1) * 0.01
        self.coords = np.asarray(load_file(s.Coord), dtype=float)
        self.basin_ids = np.asarray(load_file(s.BasinIDs, 1)).reshape(-1).astype(int)
        self.latitude = np.copy(self.coords[:, 2])
        names = getattr(s, 'BasinNames', None)                   # one basin name per line (data_load.py:57, :366-368)
        if names and os.path.isfile(names):
            with open(names) as fh:
                self.basin_names = np.array(fh.read().splitlines())
        else:
            self.basin_names = np.array(['basin_{}'.format(k) for k in range(1, s.n_basins + 1)])
        # GCAM region and country maps with their names (data_load.py:59-69, :275-286): only read by the spatial
        # aggregation of the writer, so they are loaded when present and demanded only if that aggregation is on
        self.region_ids = self.region_names = self.country_ids = self.country_names = None
        rid, rnm = getattr(s, 'GCAMRegionIDs', None), getattr(s, 'GCAMRegionNames', None)
        if _present(rid) and _present(rnm):
            self.region_ids = np.asarray(load_file(rid, 1)).reshape(-1).astype(int)
            with open(rnm) as fh:
                fh.readline()
                self.region_names = np.array([ln.split(',')[0] for ln in fh.read().split('\n') if ln != ''])
        cid, cnm = getattr(s, 'CountryIDs', None), getattr(s, 'CountryNames', None)
        if _present(cid) and _present(cnm):
            self.country_ids = np.asarray(load_file(cid, 1)).reshape(-1).astype(int)
            with open(cnm) as fh:
                self.country_names = np.array([ln.split(',')[1] for ln in fh.read().splitlines()])
        if getattr(s, 'AggregateRunoffGCAMRegion', 0) and self.region_ids is None:
            raise ValidationException('AggregateRunoffGCAMRegion = 1 needs region32_grids.csv and Rgn32Names.csv in '
                                      'the reference directory')
        if getattr(s, 'AggregateRunoffCountry', 0) and self.country_ids is None:
            raise ValidationException('AggregateRunoffCountry = 1 needs country.csv and country-names.csv in the '
                                      'reference directory')

        if s.pet_module == 'pm':
            et = np.asarray(load_file(s.pm_params), dtype=float)
            (self.cL, self.beta, self.rslimit, self.ae, self.be, self.Tminopen, self.Tminclose, self.VPDclose,
             self.VPDopen, self.RBLmin, self.RBLmax, self.rc, self.emiss) = (et[:, k] for k in range(13))
            self.alpha = np.asarray(load_file(s.pm_alpha), dtype=float)
            self.lai = np.asarray(load_file(s.pm_lai), dtype=float)
            self.laimax = np.asarray(load_file(s.pm_laimax), dtype=float)
            self.laimin = np.asarray(load_file(s.pm_laimin), dtype=float)
            self.tair_load = self.load_to_array(s.pm_tas, 'pm_tas', nan_to_num=True)
            self.TMIN_load = self.load_to_array(s.pm_tmin, 'pm_tmin', nan_to_num=True)
            self.rhs_load = self.load_to_array(s.pm_rhs, 'pm_rhs', nan_to_num=True)
            self.wind_load = self.load_to_array(s.pm_wind, 'pm_wind', nan_to_num=True)
            self.rsds_load = self.load_to_array(s.pm_rsds, 'pm_rsds', nan_to_num=True)
            self.rlds_load = self.load_to_array(s.pm_rlds, 'pm_rlds', nan_to_num=True)
            self._tairprev = None        # tairprev_load: built on first use (the device pipeline derives it in HBM)
            self.lct_load = np.nan_to_num(load_file(s.pm_lct))
            self.elev = np.nan_to_num(load_file(s.pm_elev))
        elif s.pet_module == 'none':
            self.pet_out = self.load_to_array(s.pet_file, 'pet_file')

        if s.runoff_module == 'abcd':
            self.precip = self.load_to_array(s.PrecipitationFile, 'PrecipitationFile', key=getattr(s, 'PrecipVarName', None))
            self.tmin = None if s.TempMinFile is None else self.load_to_array(
                s.TempMinFile, 'TempMinFile', nan_to_num=True, key=getattr(s, 'TempMinVarName', None))

        if s.routing_module == 'mrtm':
            self.flow_dist = self.load_routing_data(s.flow_distance, rep_val=1000)
            self.flow_dir = self.load_routing_data(s.flow_direction)
            self.str_velocity = self.load_routing_data(s.strm_veloc, rep_val=0)
            self.instream_flow = np.zeros((s.ncell,), dtype=float)
            self.chs_prev = self.load_chs_data()

        if s.calibrate:
            self.cal_obs = np.asarray(load_file(s.cal_observed, 0))[:, [0, 3]]

    @property
    def tairprev_load(self):
        """Previous-row air temperature (data_load.py:127-128: zeros_like, then rows 1.. = rows ..-1 of tair_load)."""
        if self._tairprev is None:
            self._tairprev = np.zeros(self.tair_load.shape)
            self._tairprev[1:, :] = np.nan_to_num(self.tair_load[:-1, :])
        return self._tairprev

    @tairprev_load.setter
    def tairprev_load(self, v):
        self._tairprev = v

    def load_chs_data(self):
        """Initial channel storage (data_load.py:427-438): zeros in historic mode; in future mode the last column of the
        historical run's channel storage file."""
        s = self.s
        f = getattr(s, 'ChStorageFile', None)
        if str(getattr(s, 'HistFlag', 'True')) == 'True' or f is None:
            return np.zeros((s.ncell,), dtype=float)
        arr = np.asarray(load_file(f, 0, getattr(s, 'ChStorageVarName', None)), dtype=float)
        if arr.ndim == 1:
            arr = arr[:, None]
        if arr.shape[0] != s.ncell:
            raise ValidationException('ChStorageFile has {} cells, expected {}'.format(arr.shape[0], s.ncell))
        return np.ascontiguousarray(arr[:, -1])

    def load_to_array(self, f, var_name, nan_to_num=False, key=None):
        # the big forcing files stay on disk as read-only memory maps (mmap_inputs = False restores host arrays)
        lazy = getattr(self.s, 'device_transforms', True) and getattr(self.s, 'mmap_inputs', True)
        arr = np.asanyarray(load_file(f, key=key, mmap=lazy), dtype=float)      # a float64 memory map stays one
        # np.nan_to_num of the big forcing arrays (data_load.py:120-125, :194-195) is applied on the device right after
        # the upload (xh_nan_to_num) unless device_transforms is switched off: a host pass over 2.6 GB costs seconds
        if nan_to_num and not getattr(self.s, 'device_transforms', True):
            arr = np.nan_to_num(arr)
        if arr.shape[0] != self.s.ncell or arr.shape[1] != self.s.nmonths:
            raise ValidationException('Error: Inconsistent {0} data grid size. Expecting size: {1}. Received size: {2}'
                                      .format(var_name, (self.s.ncell, self.s.nmonths), arr.shape))
        return arr

    def load_routing_data(self, fn, rep_val=None):
        """Per-cell vector from a 1-D array or a 2-D DRT map (data_load.py:392-413)."""
        fd = np.asarray(load_file(fn), dtype=float)
        if fd.ndim == 2 and fd.shape[1] == self.s.ngridcol:
            r = self.coords[:, 4].astype(int) - 1
            c = self.coords[:, 3].astype(int) - 1
            map_index = np.ravel_multi_index((r, c), (self.s.ngridrow, self.s.ngridcol), order='F')
            # the DRT maps cover 280 of the 360 rows and sit 68 rows up from the bottom (data_load.py:392, skip=68)
            skip = 68 if (fd.shape[0] == 280 and self.s.ngridrow == 360) else self.s.ngridrow - fd.shape[0]
            v = vectorize(fd, self.s.ngridrow, self.s.ngridcol, map_index, skip=skip)
        else:
            v = fd.reshape(-1).copy()
        if v.shape[0] != self.s.ncell:
            raise ValidationException('routing input {} has {} cells, expected {}'.format(fn, v.shape[0], self.s.ncell))
        if rep_val is not None:
            v[v < rep_val] = rep_val
        return v
