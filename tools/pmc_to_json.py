"""Turn two rocprofv3 --pmc runs (FETCH_SIZE, WRITE_SIZE; separate passes) into profiles/<round>/pmc_traffic.json.

HBM bytes per launch = FETCH_SIZE x 1024 x k_r + WRITE_SIZE x 1024 x k_w.  k_r = 2 for kernels whose reads are wide
coalesced streams (gfx950 counts 128-B requests as 64 B: MI355X_MICROARCH.md, HBM section), k_w = 1 for 16-byte streaming
stores.  For the routing kernel the factors are CALIBRATED: tools/micro/sc1_traffic.hip moves a known number of bytes with
the streams' own access shapes (raw-buffer 16-byte sc1 stores / loads, eight whole lines per instruction); pass the two
calibration runs and the byte count as extra arguments and the measured factors are applied (and recorded) instead of 1.

usage: pmc_to_json.py <fetch_dir> <write_dir> <out.json> [<calib_fetch_dir> <calib_write_dir> <bytes_per_launch>]
"""
import collections, csv, glob, json, sys

fetch_dir, write_dir, out = sys.argv[1:4]
WIDE = {'k_pm_pet': 2.0, 'k_synth': 2.0, 'k_abcd_tile<false': 2.0, 'k_abcd_tile<true': 2.0}      # whole-line streams
KEYS = ('k_pm_pet', 'k_abcd_tile<false', 'k_abcd_tile<true', 'k_abcd<true>', 'k_abcd<false>', 'k_abcd_basin_mean',
        'k_mrtm_wave_args', 'k_mrtm_rsum', 'k_mrtm_wave', 'k_mrtm_skew', 'k_mrtm_flow', 'k_mrtm_units', 'k_synth', 'k_sc1_store',
        'k_sc1_load', 'k_plain_store', 'k_plain_load')


def read(d):
    agg = collections.defaultdict(list)
    for f in glob.glob(d + '/*/*counter_collection.csv'):
        for r in csv.DictReader(open(f)):
            name = r['Kernel_Name']
            key = next((k for k in KEYS if k in name), None)
            if key:
                agg[key].append(float(r['Counter_Value']))
    return {k: sum(v) / len(v) for k, v in agg.items()}


f, w = read(fetch_dir), read(write_dir)
calib = None
if len(sys.argv) >= 7:
    cf, cw, nbytes = read(sys.argv[4]), read(sys.argv[5]), float(sys.argv[6])
    calib = {'bytes_per_launch': nbytes,
             'sc1_load_FETCH_SIZE_KB': cf.get('k_sc1_load'), 'sc1_store_WRITE_SIZE_KB': cw.get('k_sc1_store'),
             'plain_load_FETCH_SIZE_KB': cf.get('k_plain_load'), 'plain_store_WRITE_SIZE_KB': cw.get('k_plain_store'),
             'sc1_store_FETCH_SIZE_KB': cf.get('k_sc1_store'), 'sc1_load_WRITE_SIZE_KB': cw.get('k_sc1_load')}
    calib['fetch_factor_sc1'] = nbytes / (cf['k_sc1_load'] * 1024) if cf.get('k_sc1_load') else None
    calib['write_factor_sc1'] = nbytes / (cw['k_sc1_store'] * 1024) if cw.get('k_sc1_store') else None
    calib['fetch_factor_plain'] = nbytes / (cf['k_plain_load'] * 1024) if cf.get('k_plain_load') else None
    calib['write_factor_plain'] = nbytes / (cw['k_plain_store'] * 1024) if cw.get('k_plain_store') else None
res = {}
for k in sorted(set(f) | set(w)):
    if k.startswith('k_sc1') or k.startswith('k_plain') or k == 'k_mrtm_wave_args':
        continue
    kr, kw, note = WIDE.get(k, 1.0), 1.0, None
    if k in WIDE:
        note = 'reads are wide coalesced streams: FETCH_SIZE doubled'
    elif k in ('k_mrtm_rsum', 'k_mrtm_wave', 'k_mrtm_skew') and calib and calib['fetch_factor_sc1'] and calib['write_factor_sc1']:
        kr, kw = calib['fetch_factor_sc1'], calib['write_factor_sc1']
        note = ('stream traffic dominates: factors measured with tools/micro/sc1_traffic.hip on the same access shapes '
                '(16-byte sc1 raw-buffer loads / stores, whole lines)')
    else:
        note = 'narrow / scattered reads: FETCH_SIZE uncalibrated, taken as is'
    res[k] = {'FETCH_SIZE_KB': f.get(k), 'WRITE_SIZE_KB': w.get(k), 'fetch_factor': kr, 'write_factor': kw,
              'hbm_bytes': (f.get(k, 0) * kr + w.get(k, 0) * kw) * 1024, 'note': note}
if calib:
    res['_calibration'] = calib
import datetime
res['_meta'] = {'collected': datetime.datetime.now().strftime('%Y-%m-%d %H:%M'),
                'command': 'rocprofv3 --pmc FETCH_SIZE | WRITE_SIZE (separate passes) -- python3 bench.py --steps 2 --warmup 1 '
                           '--no-cpu-baseline --no-end-to-end',
                'kernels': sorted(k for k in res if not k.startswith('_'))}
json.dump(res, open(out, 'w'), indent=1)
print(json.dumps(res, indent=1))
