"""Turn rocprofv3 --pmc instruction / cycle passes into profiles/<round>/pmc_insts.json (read by bench.py for the
fp64-VALU fraction of Penman-Monteith): per device kernel, the mean per launch of every counter collected, plus the
date of the passes and the kernels seen.

usage: pmc_insts_json.py <out.json> <pass_dir> [<pass_dir> ...]
"""
import collections
import csv
import datetime
import glob
import json
import sys

KEYS = ('k_pm_pet', 'k_pm_pressure', 'k_abcd_tile<false', 'k_abcd_tile<true', 'k_abcd<true>', 'k_abcd<false>',
        'k_abcd_basin_mean', 'k_mrtm_wave_args', 'k_mrtm_rsum', 'k_mrtm_wave', 'k_mrtm_skew', 'k_mrtm_flow', 'k_mrtm_units',
        'k_calib_march_m<true>', 'k_calib_march_m<false>', 'k_calib_kge_m', 'k_calib_series_m', 'k_calib_de_step', 'k_calib_split')
out, dirs = sys.argv[1], sys.argv[2:]
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for d in dirs:
    for f in glob.glob(d + '/*/*counter_collection.csv'):
        for r in csv.DictReader(open(f)):
            key = next((k for k in KEYS if k in r['Kernel_Name']), None)
            if key:
                agg[key][r['Counter_Name']].append(float(r['Counter_Value']))
res = {k: {c: sum(v) / len(v) for c, v in cs.items()} for k, cs in agg.items()}
for k, cs in agg.items():
    res[k]['launches'] = max(len(v) for v in cs.values())
res['_meta'] = {'collected': datetime.datetime.now().strftime('%Y-%m-%d %H:%M'),
                'command': 'rocprofv3 --pmc <counters> (own passes, no trace) -- python3 bench.py --steps 2 --warmup 1 --order staged '
                           '--no-cpu-baseline --no-end-to-end --no-secondary; the k_calib_* kernels: -- python3 bench.py --workload calib '
                           '--steps 2 --warmup 1 --no-cpu-baseline',
                'kernels': sorted(agg)}
json.dump(res, open(out, 'w'), indent=1)
print(json.dumps(res, indent=1))
