"""Turn two rocprofv3 --pmc runs (FETCH_SIZE, WRITE_SIZE; separate passes) into profiles/<round>/pmc_traffic.json.

HBM bytes per launch = FETCH_SIZE x 1024 x k + WRITE_SIZE x 1024, with k = 2 for kernels whose reads are wide
coalesced streams (gfx950 counts 128-B requests as 64 B: MI355X_MICROARCH.md, HBM section) and k = 1 (uncalibrated)
otherwise.  WRITE_SIZE is exact for 16-B streaming stores (calibrated here on k_synth: 8 arrays x 323.6 MB).
"""
import collections, csv, glob, json, sys

fetch_dir, write_dir, out = sys.argv[1:4]
WIDE = {'k_pm_pet': 2.0, 'k_synth': 2.0, 'k_abcd_tile<false': 2.0, 'k_abcd_tile<true': 2.0}      # whole-line streams


def read(d):
    agg = collections.defaultdict(list)
    for f in glob.glob(d + '/*/*counter_collection.csv'):
        for r in csv.DictReader(open(f)):
            name = r['Kernel_Name']
            key = next((k for k in ('k_pm_pet', 'k_abcd_tile<false', 'k_abcd_tile<true', 'k_abcd<true>', 'k_abcd<false>', 'k_abcd_basin_mean',
                                    'k_mrtm_skew', 'k_mrtm_flow',
                                    'k_mrtm_units', 'k_synth') if k in name), None)
            if key:
                agg[key].append(float(r['Counter_Value']))
    return {k: sum(v) / len(v) for k, v in agg.items()}


f, w = read(fetch_dir), read(write_dir)
res = {}
for k in sorted(set(f) | set(w)):
    kk = WIDE.get(k, 1.0)
    res[k] = {'FETCH_SIZE_KB': f.get(k), 'WRITE_SIZE_KB': w.get(k), 'fetch_factor': kk,
              'hbm_bytes': (f.get(k, 0) * kk + w.get(k, 0)) * 1024,
              'note': 'reads are wide coalesced streams: FETCH_SIZE doubled' if kk == 2 else
                      'narrow / scattered reads: FETCH_SIZE uncalibrated, taken as is'}
json.dump(res, open(out, 'w'), indent=1)
print(json.dumps(res, indent=1))
