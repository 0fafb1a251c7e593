"""Deterministic synthetic 0.5-degree world and climate forcing.

The reference's input data (Zenodo archive; xanthos/install_supplement.py:19) is not available, so benchmarks
and tests run on a synthetic world with the same shapes and value ranges as the `pm_abcd_mrtm` example:
67,420 land cells on a 360 x 720 grid, 235 basins, 8 land classes.

Every random draw is a counter-based SplitMix64 hash of (seed, stream, index), so a world or a forcing block
can be regenerated identically anywhere, in any order and in pieces (per rank, per cell range) without
depending on a numpy Generator version.
"""
import heapq
from types import SimpleNamespace

import numpy as np

MASTER_SEED = 20240807
_U64 = np.uint64
_GOLD = _U64(0x9E3779B97F4A7C15)


def splitmix64(x):
    """SplitMix64 finaliser on a uint64 array (wrapping arithmetic)."""
    with np.errstate(over='ignore'):
        z = (np.asarray(x, dtype=_U64) + _GOLD)
        z = (z ^ (z >> _U64(30))) * _U64(0xBF58476D1CE4E5B9)
        z = (z ^ (z >> _U64(27))) * _U64(0x94D049BB133111EB)
        return z ^ (z >> _U64(31))


def uniform(seed, stream, idx):
    """U[0,1) doubles for integer indices ``idx`` on random stream ``stream``."""
    with np.errstate(over='ignore'):
        key = splitmix64(_U64(seed) * _U64(0xD1342543DE82EF95) + _U64(stream) * _U64(0xA0761D6478BD642F))
        h = splitmix64(np.asarray(idx, dtype=_U64) ^ key)
    return (h >> _U64(11)).astype(np.float64) * (1.0 / 9007199254740992.0)


def normal(seed, stream, idx):
    """Standard normal (Box-Muller on two hashed uniforms)."""
    u1 = uniform(seed, 2 * stream, idx)
    u2 = uniform(seed, 2 * stream + 1, idx)
    return np.sqrt(-2.0 * np.log(1.0 - u1)) * np.cos(2.0 * np.pi * u2)


# D8 code for a step (drow, dcol); "row + 1" is 'up' in the reference's decode (mrtm.py:236-240)
_D8 = {(0, 1): 1, (-1, 1): 2, (-1, 0): 4, (-1, -1): 8, (0, -1): 16, (1, -1): 32, (1, 0): 64, (1, 1): 128}


def make_world(nrow=360, ncol=720, ncell=67420, n_basins=235, nlcs=8, lc_years=(1970, 1990, 2005),
               seed=MASTER_SEED, row_margin=None, outlet_frac=0.03):
    """Grow ``n_basins`` tree-structured basins (Eden growth from outlet seeds) until ``ncell`` land cells exist.

    Returns a namespace with the arrays the reference's DataLoader would hold (data_load.py:35-224):
    coords [ncell,5] (id, lon, lat, ilon, ilat; 1-based indices), basin_ids [ncell] (1..n_basins),
    flow_dir [ncell] D8 codes (0 at outlets), area km2, flow_dist m, velocity m/s, elev [ncell,1],
    lct [ncell, nlcs, len(lc_years)], Penman-Monteith tables, ABCD parameters [n_basins,5].
    """
    if row_margin is None:
        row_margin = nrow // 12
    r_lo, r_hi = row_margin, nrow - row_margin
    usable = (r_hi - r_lo) * ncol
    if ncell > usable:
        raise ValueError('ncell does not fit the grid')

    # ---- basin seeds and growth weights (log-normal => heavy-tailed basin sizes)
    sidx = np.arange(n_basins)
    seed_r = r_lo + (uniform(seed, 1, sidx) * (r_hi - r_lo)).astype(int)
    seed_c = (uniform(seed, 2, sidx) * ncol).astype(int)
    weight = np.exp(0.45 * normal(seed, 3, sidx))

    owner = np.zeros((nrow, ncol), dtype=np.int32)          # basin id, 0 = ocean
    code = np.zeros((nrow, ncol), dtype=np.int32)           # D8 code towards the parent
    order = []                                              # growth order of (row, col)
    heap = []
    for b in range(n_basins):
        heapq.heappush(heap, (0.0, int(seed_r[b]), int(seed_c[b]), b + 1, 0))
    steps = ((-1, -1), (-1, 0), (-1, 1), (0, -1), (0, 1), (1, -1), (1, 0), (1, 1))
    # pre-hashed exponential waiting times per (cell, direction)
    wait = -np.log(1.0 - uniform(seed, 4, np.arange(nrow * ncol * 8))).reshape(nrow, ncol, 8)
    while heap and len(order) < ncell:
        t, r, c, b, d8 = heapq.heappop(heap)
        if owner[r, c]:
            continue
        owner[r, c] = b
        code[r, c] = d8
        order.append((r, c))
        wb = weight[b - 1]
        for k, (dr, dc) in enumerate(steps):
            rr, cc = r + dr, c + dc
            if r_lo <= rr < r_hi and 0 <= cc < ncol and not owner[rr, cc]:
                # the child at (rr,cc) drains back to (r,c): step (-dr,-dc)
                heapq.heappush(heap, (t + wait[r, c, k] / wb, rr, cc, b, _D8[(-dr, -dc)]))
    if len(order) != ncell:
        raise RuntimeError('world growth stalled')

    # ---- flatten in column-major grid order like the reference's coordinate table (id increases with lon, then lat)
    rows = np.array([rc[0] for rc in order])
    cols = np.array([rc[1] for rc in order])
    perm = np.lexsort((rows, cols))
    rows, cols = rows[perm], cols[perm]
    ids = np.arange(1, ncell + 1)
    lat = -90.0 + (rows + 0.5) * (180.0 / nrow)
    lon = -180.0 + (cols + 0.5) * (360.0 / ncol)
    coords = np.stack([ids, lon, lat, cols + 1, rows + 1], axis=1).astype(float)
    basin_ids = owner[rows, cols].astype(int)
    flow_dir = code[rows, cols].astype(float)

    cidx = np.arange(ncell)
    # a few percent of cells are extra sea outlets => many small river networks inside each basin
    flow_dir[uniform(seed, 5, cidx) < outlet_frac] = 0.0

    w = SimpleNamespace()
    w.nrow, w.ncol, w.ncell, w.n_basins, w.nlcs = nrow, ncol, ncell, n_basins, nlcs
    w.lc_years = list(lc_years)
    w.coords, w.basin_ids, w.flow_dir = coords, basin_ids, flow_dir
    w.area = 3091.0 * np.cos(np.radians(lat))                                  # km2 (data_load.py:48 gives km2)
    w.flow_dist = 25e3 + 50e3 * uniform(seed, 6, cidx)
    w.flow_dist[uniform(seed, 7, cidx) < 0.01] = 1000.0                        # clamped short reaches (data_load.py:204)
    w.velocity = 0.1 + 2.4 * uniform(seed, 8, cidx)
    w.velocity[uniform(seed, 9, cidx) < 0.005] = 0.0                           # data_load.py:207
    w.elev = (3000.0 * uniform(seed, 10, cidx))[:, None]
    w.latitude = lat

    nly = len(lc_years)
    e = -np.log(1.0 - uniform(seed, 11, np.arange(ncell * nlcs * nly))).reshape(ncell, nlcs, nly)
    lct = 100.0 * e / e.sum(axis=1, keepdims=True)
    lct[uniform(seed, 12, cidx) < 0.01] = 0.0                                  # cells with no land cover at all
    w.lct = lct

    li = np.arange(nlcs)
    ti = np.arange(nlcs * 12)
    u = lambda s, i=li: uniform(seed, s, i)
    w.cL = 0.0013 + 0.0052 * u(20)
    w.beta = 100.0 + 400.0 * u(21)
    w.rslimit = 500.0 + 1500.0 * u(22)
    w.ae = 0.34 + 0.1 * u(23)
    w.be = -0.14 - 0.11 * u(24)
    w.Tminopen = 8.0 + 4.0 * u(25)
    w.Tminclose = -8.0 + 2.0 * u(26)
    w.VPDclose = 25.0 + 18.0 * u(27)
    w.VPDopen = 6.5 + 3.5 * u(28)
    w.RBLmin = 20.0 + 45.0 * u(29)
    w.RBLmax = w.RBLmin + 25.0 + 10.0 * u(30)
    w.rc = 20.0 + 100.0 * u(31)
    w.emiss = 0.94 + 0.05 * u(32)
    w.alpha = (0.05 + 0.35 * u(33, ti)).reshape(nlcs, 12)
    w.lai = (6.0 * u(34, ti)).reshape(nlcs, 12)
    w.lai[min(3, nlcs - 1)] = 0.0                                               # a bare class: lai = laimin = laimax = 0
    w.laimin = np.repeat(w.lai.min(axis=1, keepdims=True), 12, axis=1)
    w.laimax = np.repeat(w.lai.max(axis=1, keepdims=True), 12, axis=1)

    bi = np.arange(n_basins)
    w.abcd_pars = np.stack([0.9 + 0.099 * uniform(seed, 40, bi), 0.1 + 1.9 * uniform(seed, 41, bi),
                            0.01 + 0.89 * uniform(seed, 42, bi), 0.01 + 0.89 * uniform(seed, 43, bi),
                            0.1 + 0.8 * uniform(seed, 44, bi)], axis=1)
    w.seed = seed
    return w


FORCING_NAMES = ('tas', 'tmin', 'rhs', 'wind', 'rsds', 'rlds', 'precip', 'abcd_tmin')


def make_forcing(world, nmonths, seed=None, cells=None, chunk=None, nan_precip=True):
    """Monthly climate forcing [ncell(or len(cells)), nmonths] float64 (SURVEY.md section 8(d) distributions).

    ``cells``: optional array of 0-based global cell indices to generate (rank shards, samples) -- values are
    identical to the same rows of the full array.
    """
    seed = world.seed + 1 if seed is None else seed
    if chunk is None:
        chunk = max(1, 32768 // nmonths)      # keep temporaries small enough to stay in the malloc heap
    sel = np.arange(world.ncell) if cells is None else np.asarray(cells)
    n = len(sel)
    out = {k: np.empty((n, nmonths)) for k in FORCING_NAMES}
    mth = np.arange(nmonths)
    season = np.sin(2.0 * np.pi * ((mth % 12) - 3.5) / 12.0)
    for s in range(0, n, chunk):
        c = sel[s:s + chunk]
        idx = (c[:, None].astype(np.int64) * 4096 + mth[None, :]).astype(np.uint64)   # < 4096 months supported
        coslat = np.cos(np.radians(world.latitude[c]))[:, None]
        hemi = np.sign(world.latitude[c])[:, None]
        tas = -10.0 + 35.0 * coslat ** 2 + 10.0 * hemi * season[None, :] + 2.0 * normal(seed, 1, idx)
        tmin = tas - (2.0 + 8.0 * uniform(seed, 4, idx))
        rhs = np.clip(65.0 + 20.0 * normal(seed, 3, idx), 5.0, 100.0)
        out['tas'][s:s + chunk] = tas
        out['tmin'][s:s + chunk] = tmin
        out['rhs'][s:s + chunk] = rhs
        out['wind'][s:s + chunk] = 0.5 + 7.5 * uniform(seed, 10, idx)
        out['rsds'][s:s + chunk] = 30.0 + 300.0 * uniform(seed, 11, idx)
        out['rlds'][s:s + chunk] = 150.0 + 280.0 * uniform(seed, 12, idx)
        pr = -40.0 * (np.log(1.0 - uniform(seed, 13, idx)) + np.log(1.0 - uniform(seed, 14, idx)))
        if nan_precip:
            pr[uniform(seed, 15, c) < 0.001] = np.nan                                # missing-data cells (kept NaN)
        out['precip'][s:s + chunk] = pr
        out['abcd_tmin'][s:s + chunk] = tmin
    return out


def data_bag(world, forcing):
    """Attribute bag with the field names run_pmpet reads from the reference DataLoader (data_load.py:92-135)."""
    d = SimpleNamespace()
    for k in ('cL', 'beta', 'rslimit', 'ae', 'be', 'Tminopen', 'Tminclose', 'VPDclose', 'VPDopen', 'RBLmin',
              'RBLmax', 'rc', 'emiss', 'alpha', 'lai', 'laimax', 'laimin', 'elev'):
        setattr(d, k, getattr(world, k))
    d.tair_load = forcing['tas']
    d.TMIN_load = forcing['tmin']
    d.rhs_load = forcing['rhs']
    d.wind_load = forcing['wind']
    d.rsds_load = forcing['rsds']
    d.rlds_load = forcing['rlds']
    d.tairprev_load = np.zeros_like(d.tair_load)
    d.tairprev_load[1:, :] = d.tair_load[:-1, :]             # the reference's cell-shift (data_load.py:128-129)
    d.lct_load = world.lct
    return d


def write_example(root, world, forcing, start_year, end_year, project='pm_abcd_mrtm_synth', runoff_spinup=36,
                  routing_spinup=None, output_vars=('q', 'avgchflow'), obs=None, post=False, aggregates=False,
                  hist_flag=True, ch_storage=None, output_format=1, output_in_year=0):
    """Write ``world`` + ``forcing`` as a Xanthos-style input tree under ``root`` and return the .ini path.

    Layout and file names follow the reference's example (ini_reader.py:254-279, 353-381, 399-416, 425-437):
    input/reference/{Grid_Areas_ID.csv, coordinates.csv, basin.csv}, input/pet/penman_monteith/*.npy + gcam_*.csv,
    input/runoff/abcd/{pars.npy, pr.npy, tmin.npy}, input/routing/mrtm/{velocity.npy, flow_dist.npy, flow_dir.npy}.
    ``post=True`` also switches on the drought-threshold and accessible-water post-processors (basin names, reservoir
    capacity and base-flow index tables under input/reference and input/accessible; ini_reader.py:460-486).
    ``aggregates=True`` writes GCAM-region and country maps with their name tables (7 regions / 10 countries, the last
    name of each without cells; countries numbered from 0) and switches the three runoff aggregations on.
    ``hist_flag=False`` + ``ch_storage`` (array [ncell]): future mode starting from a saved channel storage file
    (ini_reader.py HistFlag / ChStorageFile, data_load.py:427-438).
    """
    import os
    inp = os.path.join(root, 'input')
    dirs = {k: os.path.join(inp, *v) for k, v in dict(ref=('reference',), pet=('pet', 'penman_monteith'),
                                                      ro=('runoff', 'abcd'), rt=('routing', 'mrtm')).items()}
    for d in dirs.values():
        os.makedirs(d, exist_ok=True)
    np.savetxt(os.path.join(dirs['ref'], 'Grid_Areas_ID.csv'), world.area * 100.0, delimiter=',', fmt='%.17g')   # ha
    np.savetxt(os.path.join(dirs['ref'], 'coordinates.csv'), world.coords, delimiter=',', fmt='%.17g')
    np.savetxt(os.path.join(dirs['ref'], 'basin.csv'), np.concatenate([[0], world.basin_ids]), fmt='%d')   # 1 header row
    et = np.stack([world.cL, world.beta, world.rslimit, world.ae, world.be, world.Tminopen, world.Tminclose,
                   world.VPDclose, world.VPDopen, world.RBLmin, world.RBLmax, world.rc, world.emiss], axis=1)
    np.savetxt(os.path.join(dirs['pet'], 'gcam_ET_para.csv'), et, delimiter=',', fmt='%.17g')
    for name, arr in (('gcam_albedo', world.alpha), ('gcam_lai', world.lai), ('gcam_laimin', world.laimin),
                      ('gcam_laimax', world.laimax)):
        np.savetxt(os.path.join(dirs['pet'], name + '.csv'), arr, delimiter=',', fmt='%.17g')
    np.save(os.path.join(dirs['pet'], 'elev.npy'), world.elev)
    np.save(os.path.join(dirs['pet'], 'lct.npy'), world.lct)
    for key in ('tas', 'tmin', 'rhs', 'wind', 'rsds', 'rlds'):
        np.save(os.path.join(dirs['pet'], key + '.npy'), forcing[key])
    np.save(os.path.join(dirs['ro'], 'pars.npy'), world.abcd_pars)
    np.save(os.path.join(dirs['ro'], 'pr.npy'), forcing['precip'])
    np.save(os.path.join(dirs['ro'], 'tmin.npy'), forcing['abcd_tmin'])
    np.save(os.path.join(dirs['rt'], 'velocity.npy'), world.velocity)
    np.save(os.path.join(dirs['rt'], 'flow_dist.npy'), world.flow_dist)
    np.save(os.path.join(dirs['rt'], 'flow_dir.npy'), world.flow_dir)
    nmonths = (end_year - start_year + 1) * 12
    ini = os.path.join(root, project + '.ini')
    calib = ''
    if obs is not None:
        obs_file = os.path.join(inp, 'obs.csv')
        np.savetxt(obs_file, obs, delimiter=',', fmt='%.17g')
        calib = ('\n[Calibrate]\nset_calibrate = 0\nobserved = {}\nobs_unit = km3_per_mth\ncalib_out_dir = {}\n'
                 'calibration_basins = 1-2\n').format(obs_file, os.path.join(root, 'calib_out'))
    post_project = post_sections = ''
    if post:
        acc = os.path.join(inp, 'accessible')
        os.makedirs(acc, exist_ok=True)
        nb = world.n_basins
        with open(os.path.join(dirs['ref'], 'BasinNames235.txt'), 'w') as fh:
            fh.write('\n'.join('Basin {:03d}'.format(k) for k in range(1, nb + 1)) + '\n')
        cap = uniform(splitmix64(np.arange(nb, dtype=np.uint64) + np.uint64(77001)), 0.0, 5.0)
        bfi = uniform(splitmix64(np.arange(nb, dtype=np.uint64) + np.uint64(77002)), 0.2, 0.9)
        np.savetxt(os.path.join(acc, 'total_reservoir_storage.csv'), cap, fmt='%.17g')
        with open(os.path.join(acc, 'bfi_per_basin.csv'), 'w') as fh:
            fh.write('basin_id,bfi_avg\n' + ''.join('{},{!r}\n'.format(k + 1, float(bfi[k])) for k in range(nb)))
        post_project = 'AccWatDir = accessible\nCalculateDroughtStats = 1\nCalculateAccessibleWater = 1\n'
        post_sections = ('\n[Drought]\ndrought_var = q\nthreshold_nper = 12\nthreshold_start_year = {y0}\n'
                         'threshold_end_year = {y1}\n\n[AccessibleWater]\nResCapacityFile = total_reservoir_storage.csv\n'
                         'BfiFile = bfi_per_basin.csv\nHistEndYear = {y1}\nGCAM_StartYear = {y0}\nGCAM_EndYear = {y1}\n'
                         'GCAM_YearStep = 1\nMovingMeanWindow = 3\nEnv_FlowPercent = 0.1\n').format(y0=start_year, y1=end_year)
    if aggregates:
        names = os.path.join(dirs['ref'], 'BasinNames235.txt')
        if not os.path.isfile(names):
            with open(names, 'w') as fh:
                fh.write('\n'.join('Basin {:03d}'.format(k) for k in range(1, world.n_basins + 1)) + '\n')
        np.savetxt(os.path.join(dirs['ref'], 'region32_grids.csv'), np.concatenate([[0], world.basin_ids % 6 + 1]), fmt='%d')
        with open(os.path.join(dirs['ref'], 'Rgn32Names.csv'), 'w') as fh:
            fh.write('region,region_id\n' + '\n'.join('Region {},{}'.format(k, k) for k in range(1, 8)))
        np.savetxt(os.path.join(dirs['ref'], 'country.csv'), np.concatenate([[0], world.basin_ids % 9]), fmt='%d')
        with open(os.path.join(dirs['ref'], 'country-names.csv'), 'w') as fh:
            fh.write(''.join('{},Country {}\n'.format(k, k) for k in range(10)))
        post_project += 'AggregateRunoffBasin = 1\nAggregateRunoffCountry = 1\nAggregateRunoffGCAMRegion = 1\n'
    chs_lines = ''
    if not hist_flag:
        np.save(os.path.join(dirs['rt'], 'ch_storage.npy'), np.asarray(ch_storage, dtype=float))
        chs_lines = 'ChStorageFile = {}\nChStorageVarName = chs\n'.format(os.path.join(dirs['rt'], 'ch_storage.npy'))
    with open(ini, 'w') as fh:
        fh.write('''[Project]
# synthetic pm_abcd_mrtm example written by xanthos_amd.synth.write_example
ProjectName = {project}
RootDir = {root}
InputFolder = input
OutputFolder = output
RefDir = reference
pet_dir = pet
RunoffDir = runoff
RoutingDir = routing
HistFlag = {hist}
n_basins = {nb}
ncell = {ncell}
ngridrow = {nrow}
ngridcol = {ncol}
StartYear = {y0}
EndYear = {y1}
output_vars = {ov}
OutputFormat = {ofmt}
OutputUnit = 0
OutputInYear = {oyear}
Calibrate = {cal}
{post_project}
[PET]
pet_module = pm
[[penman-monteith]]
pet_dir = penman_monteith
pm_tas = tas.npy
pm_tmin = tmin.npy
pm_rhs = rhs.npy
pm_rlds = rlds.npy
pm_rsds = rsds.npy
pm_wind = wind.npy
pm_lct = lct.npy
pm_nlcs = {nlcs}
pm_water_idx = 0
pm_snow_idx = 6
pm_lc_years = {lcy}

[Runoff]
runoff_module = abcd
[[abcd]]
runoff_dir = abcd
calib_file = pars.npy
runoff_spinup = {rsp}
jobs = -1
PrecipitationFile = {pr}
TempMinFile = {tn}
{chs}
[Routing]
routing_module = mrtm
[[mrtm]]
routing_dir = mrtm
routing_spinup = {rtsp}
channel_velocity = velocity.npy
flow_distance = flow_dist.npy
flow_direction = flow_dir.npy
{calib}{post_sections}'''.format(ofmt=int(output_format), oyear=int(output_in_year), chs=chs_lines, hist='True' if hist_flag else 'False', post_project=post_project, post_sections=post_sections, project=project, root=root, nb=world.n_basins, ncell=world.ncell, nrow=world.nrow, ncol=world.ncol,
                  y0=start_year, y1=end_year, ov=', '.join(output_vars), cal=int(obs is not None), nlcs=world.nlcs,
                  lcy=', '.join(str(y) for y in world.lc_years), rsp=runoff_spinup,
                  rtsp=nmonths if routing_spinup is None else routing_spinup,
                  pr=os.path.join(dirs['ro'], 'pr.npy'), tn=os.path.join(dirs['ro'], 'tmin.npy'), calib=calib))
    return ini
