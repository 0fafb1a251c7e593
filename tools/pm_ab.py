"""A/B of two builds of the library on one box, alternating: config 2 (pm_abcd) or, with a third argument `pm_abcd_mrtm`,
config 3 timings.  usage: pm_ab.py libA.so libB.so [workload]"""
import os, subprocess, sys, json
root = os.path.join(os.path.dirname(os.path.abspath(__file__)), '..')
code = ("import sys, json; sys.path.insert(0, %r); from xanthos_amd import _hip; _hip.LIB_PATH = sys.argv[1]; sys.argv = ['bench.py', '--workload', sys.argv[2], '--steps', '20', '--warmup', '5', '--no-cpu-baseline', '--no-end-to-end']; "
        "import runpy; runpy.run_path(%r, run_name='__main__')") % (root, os.path.join(root, 'bench.py'))
workload = sys.argv[3] if len(sys.argv) > 3 else 'pm_abcd'
for rep in range(3):
    for lib in sys.argv[1:3]:
        out = subprocess.run([sys.executable, '-c', code, os.path.abspath(lib), workload], capture_output=True, text=True)
        try:
            d = json.loads(out.stdout.strip().splitlines()[-1])
            print(os.path.basename(lib), 'pm_pet %.4f ms  step %.4f ms' % (d['kernels']['pm_pet']['avg_ms'], d['ms_per_step']),
                  ('mrtm_route %.4f ms' % d['kernels']['mrtm_route']['avg_ms']) if 'mrtm_route' in d['kernels'] else '')
        except Exception as e:
            print(os.path.basename(lib), 'failed', out.stderr[-500:])
