"""OutWriter for the hot-path outputs -- the array math of xanthos/data_writer/out_writer.py on the GPU.

Same constructor and methods as the reference class (:33-265): ``OutWriter(settings, grid_areas, all_outputs)``,
``write()``, ``get(var)``, ``write_aggregates(ref, values, basin, country, region)``.  Month -> year aggregation
(sum; mean for ``avgchflow``, :100-108), the mm -> km3 conversion (:111-112) and the basin / country / region sums
(:250-265) run as HIP kernels (csrc/xh_agg.hip) on arrays that may already be resident in HBM; only the (12x smaller
for yearly output) results cross PCIe.  Files are written as ``.csv`` (OutputFormat 1, the reference's layout:
an ``id`` column of 1-based cell ids and one column per time step) or ``.npy`` (4); NetCDF / MATLAB / parquet
(0, 2, 3) need pandas writers outside the hot path and raise.
"""
import logging
import os

import numpy as np

from .. import _hip

FORMAT_NETCDF, FORMAT_CSV, FORMAT_MAT, FORMAT_PARQUET, FORMAT_NPY = 0, 1, 2, 3, 4
UNIT_MM_MTH, UNIT_KM3_MTH = 0, 1
NMONTHS = 12


class OutWriter:

    def __init__(self, settings, grid_areas, all_outputs, device=None):
        self.output_names = [o for o in settings.output_vars if o in all_outputs.keys()]
        self.ctx = _hip.get_context(getattr(settings, 'device', 0) if device is None else device)
        self.inputs = {o: all_outputs[o] for o in self.output_names}           # host ndarray or DeviceArray
        self.outputs = [None] * len(self.output_names)
        self.grid_areas = np.asarray(grid_areas, dtype=np.float64)
        self.conversion_mm_km3 = self.grid_areas / 1e6
        self.proj_name = settings.ProjectName
        self.out_folder = settings.OutputFolder
        self.out_format = settings.OutputFormat
        self.out_unit = settings.OutputUnit
        self.out_unit_str = '{}per{}'.format(('mm', 'km3')[settings.OutputUnit], ('month', 'year')[settings.OutputInYear])
        self.output_in_year = settings.OutputInYear
        years = range(settings.StartYear, settings.EndYear + 1)
        self.time_steps = ([str(y) for y in years] if self.output_in_year else
                           ['{}{:02}'.format(y, m) for y in years for m in range(1, NMONTHS + 1)])
        if self.out_format not in (FORMAT_NETCDF, FORMAT_CSV, FORMAT_MAT, FORMAT_PARQUET, FORMAT_NPY):
            logging.warning('Output format {} is invalid; writing output as .csv'.format(self.out_format))
            self.out_format = FORMAT_CSV

    def get(self, varstr, host=True):
        """The written array of a variable (:77-79).  host=False may return the DeviceArray a monthly, unconverted
        variable was saved from."""
        i = self.output_names.index(varstr)
        if host and isinstance(self.outputs[i], _hip.DeviceArray):
            self.outputs[i] = self.outputs[i].download()
        return self.outputs[i]

    # ---- device helpers
    def _on_device(self, arr):
        if isinstance(arr, _hip.DeviceArray):
            return arr, False
        return self.ctx.upload(np.asarray(arr, dtype=np.float64)), True

    def agg_to_year(self, arr, func='sum', scale=None):
        """[ncell, nmonths] -> [ncell, nyears] (:237-248), optionally x scale[c] afterwards. Returns a host array."""
        return self._agg(arr, NMONTHS, 0 if func == 'sum' else 1, scale)

    def _agg(self, arr, group, mode, scale):
        src, mine = self._on_device(arr)
        ncell, ncols = src.shape
        dst = self.ctx.empty((ncell, ncols // group))
        d_scale = None if scale is None else self.ctx.upload(scale)
        self.ctx.agg_time(ncell, ncols, group, mode, d_scale, src, dst)
        out = dst.download()
        for b in (dst, d_scale, src if mine else None):
            if b is not None:
                b.free()
        return out

    def agg_spatial(self, arr, id_map, n_ids, first_id=1):
        """[ncell, t] -> [n_ids, t]: NaN-skipping sums per id (:250-265); ids first_id .. first_id + n_ids - 1."""
        src, mine = self._on_device(arr)
        ncell, ncols = src.shape
        idx = np.asarray(id_map).astype(np.int64) - first_id
        idx[(idx < 0) | (idx >= n_ids)] = -1
        dst = self.ctx.empty((n_ids, ncols))
        self.ctx.agg_spatial(ncell, ncols, n_ids, idx, src, dst)
        out = dst.download()
        dst.free()
        if mine:
            src.free()
        return out

    # ---- the reference's write() (:81-125)
    def write(self):
        if not self.output_names:
            logging.debug('No valid output variables specified')
            return
        self._npy_from_device = []                      # monthly, unconverted npy outputs still in HBM: saved side by side
        for i, var in enumerate(self.output_names):
            flow = var == 'avgchflow'
            unit = 'm3persec' if flow else self.out_unit_str
            scale = self.conversion_mm_km3 if (self.out_unit == UNIT_KM3_MTH and not flow) else None
            if self.output_in_year:
                self.outputs[i] = self._agg(self.inputs[var], NMONTHS, 1 if flow else 0, scale)
            elif scale is not None:
                self.outputs[i] = self._agg(self.inputs[var], 1, 0, scale)
            else:
                a = self.inputs[var]
                # a device array is saved from HBM (npy) or fetched for the csv writer
                keep = isinstance(a, _hip.DeviceArray) and self.out_format == FORMAT_NPY
                self.outputs[i] = a if keep else (a.download() if isinstance(a, _hip.DeviceArray) else np.asarray(a))
            filename = os.path.join(self.out_folder, '{}_{}_{}'.format(var, unit, self.proj_name))
            self.write_data(filename, var, self.outputs[i], self.time_steps, first_id=1)
        if self._npy_from_device:
            self.ctx.save_npy_many(self._npy_from_device)
            self._npy_from_device = []

    def write_aggregates(self, ref, values, basin, country, region):
        """Spatial sums of ``values`` (the written runoff) by basin / country / GCAM region (:126-158).

        As in the reference's ``agg_spatial`` (:250-265) there is one row per NAME: basins and regions are numbered from
        1 (``inc_name_idx=True``), countries from 0 (the names table keeps its 0-based index there); ids without a
        name are dropped, names without cells give NaN.  The csv carries the ``id`` and ``name`` columns the
        reference's DataFrame has."""
        filepath = os.path.join(self.out_folder, '{}_' + '{}_{}'.format(self.out_unit_str, self.proj_name))
        jobs = []
        if basin:
            names = getattr(ref, 'basin_names', None)
            n = len(names) if names is not None else getattr(ref, 'n_basin_names', int(np.max(ref.basin_ids)))
            jobs.append(('Basin_runoff', ref.basin_ids, n, 1, names))
        if country:
            if getattr(ref, 'country_ids', None) is None:
                raise ValueError('AggregateRunoffCountry needs country ids and names (country.csv, country-names.csv)')
            jobs.append(('Country_runoff', ref.country_ids, len(ref.country_names), 0, ref.country_names))
        if region:
            if getattr(ref, 'region_ids', None) is None:
                raise ValueError('AggregateRunoffGCAMRegion needs region ids and names (region32_grids.csv, '
                                 'Rgn32Names.csv)')
            jobs.append(('GCAMRegion_runoff', ref.region_ids, len(ref.region_names), 1, ref.region_names))
        out = {}
        for name, ids, n, first, names in jobs:
            logging.info('Aggregating by ' + name.split('_')[0])
            out[name] = self.agg_spatial(values, ids, n, first_id=first)
            self.write_data(filepath.format(name), name, out[name], self.time_steps, first_id=first, names=names)
        logging.info('Aggregated unit is {}'.format(self.out_unit_str))
        return out

    def write_data(self, filename, var, data, col_names, first_id=1, names=None):
        os.makedirs(self.out_folder, exist_ok=True)
        if self.out_format == FORMAT_NPY:
            if isinstance(data, _hip.DeviceArray):
                if getattr(self, '_npy_from_device', None) is not None and var in self.output_names:
                    self._npy_from_device.append((filename + '.npy', data))      # flushed at the end of write()
                else:
                    self.ctx.save_npy(filename + '.npy', data)
            else:
                np.save(filename + '.npy', data)
        elif self.out_format == FORMAT_CSV:
            ids = np.arange(first_id, first_id + data.shape[0])
            header = 'id,' + ('name,' if names is not None else '') + ','.join(col_names[:data.shape[1]])
            fmt = lambda v: '' if v != v else repr(float(v))                    # pandas writes NaN as an empty field
            with open(filename + '.csv', 'w') as fh:
                fh.write(header + '\n')
                for k, (i, row) in enumerate(zip(ids, data)):
                    label = '' if names is None else str(names[k]) + ','
                    fh.write(str(i) + ',' + label + ','.join(fmt(v) for v in row) + '\n')
        else:
            raise RuntimeError('OutputFormat {} (NetCDF / MATLAB / parquet) is written by the reference\'s pandas '
                               'writers, outside the MI355X hot path; use 1 (csv) or 4 (npy)'.format(self.out_format))
