"""Configuration reader for the pm / abcd / mrtm path -- same .ini surface as xanthos/data_reader/ini_reader.py.

The reference parses the file with ``configobj`` (not installed here) and flattens it into an attribute bag
(ini_reader.py:24-607).  This module has its own small parser for the same syntax (``[Section]``, nested
``[[subsection]]``, ``key = value``, ``#`` comments, comma lists, optional quotes) and builds the same attributes for
the sections the hot path reads: ``[Project]``, ``[PET][[penman-monteith]]``, ``[Runoff][[abcd]]``,
``[Routing][[mrtm]]`` and ``[Calibrate]``.  Selector strings are lower-cased and validated exactly like the
reference (:214, :309, :397); selectors that belong to other reference modules are rejected with a clear message
because only the MI355X hot path is implemented here.  ``update()`` keeps the in-memory override hook (:598-607).
"""
import os
import logging


class ValidationException(Exception):
    """Invalid Xanthos configuration (ini_reader.py:17)."""


def parse_ini(path):
    """Nested dict of the file's sections; values are str or list of str (comma separated)."""
    root = {}
    stack = [root]
    with open(path, 'r') as fh:
        for raw in fh:
            line = _strip_comment(raw).strip()
            if not line:
                continue
            if line.startswith('['):
                depth = len(line) - len(line.lstrip('['))
                name = line.strip('[]').strip()
                if depth < 1 or depth > len(stack):
                    raise ValidationException('bad section nesting: ' + raw.strip())
                del stack[depth:]
                sec = {}
                stack[-1][name] = sec
                stack.append(sec)
            elif '=' in line:
                key, val = line.split('=', 1)
                stack[-1][key.strip()] = _value(val.strip())
            else:
                raise ValidationException('cannot parse line: ' + raw.strip())
    return root


def _strip_comment(line):
    out, quote = [], None
    for ch in line:
        if quote:
            if ch == quote:
                quote = None
        elif ch in '"\'':
            quote = ch
        elif ch == '#':
            break
        out.append(ch)
    return ''.join(out)


def _value(text):
    parts = [p.strip().strip('"\'') for p in text.split(',')]
    if len(parts) > 1:
        return [p for p in parts if p != '']
    return parts[0] if parts else ''


class ConfigReader:
    """Attribute bag of settings for one run (ini_reader.py:24)."""

    PET_OTHER = ('hargreaves', 'hs', 'thornthwaite')
    RUNOFF_OTHER = ('gwam',)

    def __init__(self, ini):
        c = parse_ini(ini) if not isinstance(ini, dict) else ini
        try:
            p = c['Project']
        except KeyError:
            raise ValidationException('no [Project] section in ' + str(ini))

        self.root = p['RootDir']
        self.ProjectName = p['ProjectName']
        self.OutputNameStr = p['ProjectName']
        self.InputFolder = os.path.join(self.root, p['InputFolder'])
        self.OutDir = os.path.join(self.root, p['OutputFolder'])
        self.OutputFolder = os.path.join(self.OutDir, self.ProjectName)

        self.Reference = os.path.join(self.InputFolder, p['RefDir']) if 'RefDir' in p else None
        self.PET = os.path.join(self.InputFolder, p['pet_dir']) if 'pet_dir' in p else self.InputFolder
        self.RunoffDir = os.path.join(self.InputFolder, p['RunoffDir']) if 'RunoffDir' in p else self.InputFolder
        self.RoutingDir = os.path.join(self.InputFolder, p['RoutingDir']) if 'RoutingDir' in p else self.InputFolder

        # project-level settings (ini_reader.py:117-142); the grid is hard-wired in the reference (:117-119) --
        # here ncell / ngridrow / ngridcol may be overridden for reduced test grids
        self.ncell = int(p.get('ncell', 67420))
        self.ngridrow = int(p.get('ngridrow', 360))
        self.ngridcol = int(p.get('ngridcol', 720))
        self.n_basins = int(p['n_basins'])
        # spelled 'True' / 'False' in the reference's examples; its own code compares the raw string in three different ways
        # (ini_reader.py:330 == 'False', :582 .lower() in [...], data_load.py:431 == "True"), so a config with 'false' or
        # 'F' would silently run a mix of both modes there.  Normalised once here: historic iff it reads as true.
        raw_hist = str(p.get('HistFlag', 'True')).strip()
        if raw_hist.lower() not in ('true', 't', 'yes', 'y', '1', 'false', 'f', 'no', 'n', '0'):
            raise ValidationException("HistFlag must be True or False, not '{}'".format(raw_hist))
        self.historic = raw_hist.lower() in ('true', 't', 'yes', 'y', '1')
        self.HistFlag = 'True' if self.historic else 'False'
        self.StartYear = int(p['StartYear'])
        self.EndYear = int(p['EndYear'])
        ov = p.get('output_vars', '')
        self.output_vars = ov if isinstance(ov, list) else [ov]
        for key in ('OutputFormat', 'OutputUnit', 'OutputInYear', 'AggregateRunoffBasin', 'AggregateRunoffCountry',
                    'AggregateRunoffGCAMRegion', 'PerformDiagnostics', 'CreateTimeSeriesPlot', 'CalculateDroughtStats',
                    'CalculateAccessibleWater', 'CalculateHydropowerPotential', 'CalculateHydropowerActual'):
            setattr(self, key, int(p.get(key, 0)))
        self.calibrate = int(p.get('Calibrate', 0))
        self.nmonths = (self.EndYear - self.StartYear + 1) * 12
        self.device = int(p.get('device', 0))

        self.configure_pet(c.get('PET'))
        self.configure_runoff(c.get('Runoff'))
        self.configure_routing(c.get('Routing'))
        self.mod_cfg = '{0}_{1}_{2}'.format(self.pet_module, self.runoff_module, self.routing_module)
        if self.mod_cfg == 'none_none_none':
            raise ValidationException('No PET, Runoff, or Routing model selected.')
        self.configure_reference_data()
        # post-processors next to the hot path (ini_reader.py:85-98, 178-184)
        if c.get('Drought') and self.CalculateDroughtStats:
            self.configure_drought_stats(c['Drought'])
        if c.get('AccessibleWater') and self.CalculateAccessibleWater:
            if 'AccWatDir' not in p:
                raise ValidationException('CalculateAccessibleWater = 1 needs AccWatDir in [Project].')
            self.AccWatDir = os.path.join(self.InputFolder, p['AccWatDir'])
            self.configure_acc_water(c['AccessibleWater'])
        for flag in ('PerformDiagnostics', 'CreateTimeSeriesPlot', 'CalculateHydropowerPotential',
                     'CalculateHydropowerActual'):
            if getattr(self, flag):
                raise ValidationException("{} = 1: this post-processor belongs to the reference's host-side modules and "
                                          "is not part of this package.".format(flag))
        if self.calibrate:
            if 'Calibrate' not in c:
                raise ValidationException('Calibrate = 1 but no [Calibrate] section.')
            self.configure_calibration(c['Calibrate'])

    # ------------------------------------------------------------------ modules
    def configure_pet(self, cfg):
        """[PET] / [[penman-monteith]] (ini_reader.py:198-300)."""
        if not cfg:
            self.pet_module = 'none'
            return
        self.pet_module = cfg['pet_module'].lower()
        if self.pet_module == 'pm':
            m = cfg['penman-monteith']
            self.pet_dir = os.path.join(self.PET, m['pet_dir'])
            for key in ('pm_tas', 'pm_tmin', 'pm_rhs', 'pm_rlds', 'pm_rsds', 'pm_wind', 'pm_lct'):
                setattr(self, key, os.path.join(self.pet_dir, m[key]))
            self.pm_nlcs = int(m['pm_nlcs'])
            self.pm_water_idx = int(m['pm_water_idx'])
            self.pm_snow_idx = int(m['pm_snow_idx'])
            years = m['pm_lc_years']
            self.pm_lc_years = [int(i) for i in (years if isinstance(years, list) else [years])]
            self.pm_params = os.path.join(self.pet_dir, 'gcam_ET_para.csv')
            self.pm_alpha = os.path.join(self.pet_dir, 'gcam_albedo.csv')
            self.pm_lai = os.path.join(self.pet_dir, 'gcam_lai.csv')
            self.pm_laimin = os.path.join(self.pet_dir, 'gcam_laimin.csv')
            self.pm_laimax = os.path.join(self.pet_dir, 'gcam_laimax.csv')
            self.pm_elev = os.path.join(self.pet_dir, 'elev.npy')
        elif self.pet_module == 'none':
            try:
                self.pet_file = cfg['pet_file']
            except KeyError:
                raise ValidationException('USAGE: Must provide a pet_file variable in the PET config section that '
                                          'contains the full path to an input PET file if not using an existing module.')
        elif self.pet_module in self.PET_OTHER:
            raise ValidationException("PET module '{0}' belongs to the reference's CPU modules; this package implements "
                                      "the MI355X hot path only (pet_module = pm).".format(self.pet_module))
        else:
            raise ValidationException("ERROR: PET module '{0}' not found. Please check "
                                      "spelling and try again.".format(self.pet_module))

    def configure_runoff(self, cfg):
        """[Runoff] / [[abcd]] (ini_reader.py:302-388)."""
        if not cfg:
            self.runoff_module = 'none'
            return
        self.runoff_module = cfg['runoff_module'].lower()
        if self.runoff_module == 'abcd':
            m = cfg['abcd']
            self.ro_model_dir = os.path.join(self.RunoffDir, m['runoff_dir'])
            self.calib_file = os.path.join(self.ro_model_dir, m['calib_file'])
            self.runoff_spinup = int(m['runoff_spinup'])
            self.ro_jobs = int(m.get('jobs', -1))
            try:
                self.PrecipitationFile = m['PrecipitationFile']
            except KeyError:
                raise ValidationException('File path not provided for the PrecipitationFile variable in the ABCD '
                                          'runoff section of the config file.')
            self.PrecipVarName = m.get('PrecipVarName')
            self.TempMinFile = m.get('TempMinFile')
            self.TempMinVarName = m.get('TempMinVarName')
            # Future mode (HistFlag = False): channel storage at the end of the historical run.  The reference reads these
            # two keys only in its gwam section (ini_reader.py:322-338), so with abcd its loader silently starts from
            # zeros (data_load.py:427-438); here they are honoured in the abcd section as well.
            self.ChStorageFile = self.ChStorageVarName = None
            if not self.historic:
                self.ChStorageFile = m.get('ChStorageFile')
                self.ChStorageVarName = m.get('ChStorageVarName')
                if not self.ChStorageFile:
                    # the reference accepts this configuration and starts from empty channels (its abcd section never
                    # reads the key; load_chs_data, data_load.py:427-438, falls back to zeros): so does this package,
                    # loudly -- XH_STRICT_FUTURE=1 turns the warning into the error it almost always deserves
                    msg = ('HistFlag = False (future mode) without ChStorageFile (and ChStorageVarName for NetCDF) in the '
                           'runoff section: routing starts from EMPTY channels, as in the reference, instead of the '
                           'channel storage the historical run ended with.')
                    if os.environ.get('XH_STRICT_FUTURE') == '1':
                        raise ValidationException(msg)
                    logging.warning(msg)
        elif self.runoff_module == 'none':
            pass
        elif self.runoff_module in self.RUNOFF_OTHER:
            raise ValidationException("Runoff module '{0}' belongs to the reference's CPU modules; this package "
                                      "implements the MI355X hot path only (runoff_module = abcd).".format(self.runoff_module))
        else:
            raise ValidationException("ERROR: Runoff module '{0}' not found. Please check "
                                      "spelling and try again.".format(self.runoff_module))

    def configure_routing(self, cfg):
        """[Routing] / [[mrtm]] (ini_reader.py:390-423)."""
        if not cfg:
            self.routing_module = 'none'
            return
        self.routing_module = cfg['routing_module'].lower()
        if self.routing_module == 'mrtm':
            m = cfg['mrtm']
            self.rt_model_dir = os.path.join(self.RoutingDir, m['routing_dir'])
            self.strm_veloc = os.path.join(self.rt_model_dir, m['channel_velocity'])
            self.flow_distance = os.path.join(self.rt_model_dir, m['flow_distance'])
            self.flow_direction = os.path.join(self.rt_model_dir, m['flow_direction'])
            self.routing_spinup = int(m['routing_spinup']) if 'routing_spinup' in m else self.nmonths
            alt = m.get('alt_runoff')
            self.alt_runoff = None if alt in (None, 'none') else os.path.join(self.rt_model_dir, alt)
            # (not a key of the reference) which form of the routing kernel: `reassociated` -- row sums as running sums along
            # chains of lanes, equal to the reference to rounding (<= 1e-9 relative; NOT bit for bit), twice as fast --, `exact`
            # -- every sum in scipy's stored order, ChStorage / Avg_ChFlow bit-identical to the reference --, or `default`: the
            # library's, which is `reassociated` since round 5 (XH_ROUTE_REASSOC=0 in the environment makes it `exact`)
            self.routing_form = str(m.get('routing_form', 'default')).strip().lower()
            if self.routing_form not in ('default', 'reassociated', 'exact'):
                raise ValidationException("routing_form must be 'reassociated', 'exact' or 'default', not '{}'".format(
                    self.routing_form))
        elif self.routing_module == 'none':
            pass
        else:
            raise ValidationException("ERROR: Routing module '{0}' not found. Please check "
                                      "spelling and try again.".format(self.routing_module))

    def configure_reference_data(self):
        """Reference grid files (ini_reader.py:425-437); only the ones the hot path reads."""
        if self.Reference:
            self.Area = os.path.join(self.Reference, 'Grid_Areas_ID.csv')
            self.Coord = os.path.join(self.Reference, 'coordinates.csv')
            self.BasinIDs = os.path.join(self.Reference, 'basin.csv')
            self.BasinNames = os.path.join(self.Reference, 'BasinNames235.txt')
            self.GCAMRegionIDs = os.path.join(self.Reference, 'region32_grids.csv')
            self.GCAMRegionNames = os.path.join(self.Reference, 'Rgn32Names.csv')
            self.CountryIDs = os.path.join(self.Reference, 'country.csv')
            self.CountryNames = os.path.join(self.Reference, 'country-names.csv')

    def configure_drought_stats(self, cfg):
        """[Drought] (ini_reader.py:460-471)."""
        self.drought_var = cfg['drought_var']
        self.drought_thresholds = cfg.get('drought_thresholds')            # optional: thresholds file of an earlier run
        if self.drought_thresholds is None:
            self.threshold_nper = int(cfg['threshold_nper'])
            self.threshold_start_year = int(cfg['threshold_start_year'])
            self.threshold_end_year = int(cfg['threshold_end_year'])
            if self.StartYear > self.threshold_start_year or self.EndYear < self.threshold_end_year:
                raise ValidationException('Drought threshold year range is outside the output year range.')

    def configure_acc_water(self, cfg):
        """[AccessibleWater] (ini_reader.py:473-486)."""
        self.ResCapacityFile = os.path.join(self.AccWatDir, cfg['ResCapacityFile'])
        self.BfiFile = os.path.join(self.AccWatDir, cfg['BfiFile'])
        self.HistEndYear = int(cfg['HistEndYear'])
        self.GCAM_StartYear = self.ck_year(int(cfg['GCAM_StartYear']))
        self.GCAM_EndYear = int(cfg['GCAM_EndYear'])
        self.GCAM_YearStep = int(cfg['GCAM_YearStep'])
        self.MovingMeanWindow = int(cfg['MovingMeanWindow'])
        self.Env_FlowPercent = float(cfg['Env_FlowPercent'])
        if self.StartYear > self.GCAM_StartYear or self.EndYear < self.GCAM_EndYear:
            raise ValidationException('Accessible water range of GCAM years are outside the range of years in climate data.')

    def ck_year(self, yr):
        """A year inside the run (ini_reader.py:547-551)."""
        if yr < self.StartYear or yr > self.EndYear:
            raise ValidationException('Accessible water year {0} is outside the range of years in the climate data.'.format(yr))
        return yr

    def configure_calibration(self, cfg):
        """[Calibrate] (ini_reader.py:506-519)."""
        self.set_calibrate = int(cfg['set_calibrate'])
        self.cal_observed = cfg['observed']
        self.obs_unit = self.ck_obs_unit(self.set_calibrate, cfg['obs_unit'])
        self.calib_out_dir = cfg['calib_out_dir']
        basins = cfg.get('calibration_basins')
        if basins is None:
            self.cal_basins = ['1-{}'.format(self.n_basins)]
        else:
            self.cal_basins = basins if isinstance(basins, list) else [basins]

    @staticmethod
    def ck_obs_unit(set_calib, unit):
        """Units accepted for the observations (ini_reader.py:521-545)."""
        valid = ('km3_per_mth', 'mm_per_mth') if set_calib == 0 else ('m3_per_sec',)
        if unit not in valid:
            raise ValidationException("Calibration data input units '{}' not in required units '{}'".format(unit, valid))
        return unit

    def update(self, args):
        """Overwrite configuration options in memory (ini_reader.py:598-607)."""
        for k, v in args.items():
            if not hasattr(self, k):
                print('Warning: {} is not a valid parameter'.format(k))
            setattr(self, k, v)
