"""A/B of two builds of the library on one box: config 2 (pm_abcd) timings, alternating.  usage: pm_ab.py libA.so libB.so"""
import os, subprocess, sys, json
root = os.path.join(os.path.dirname(os.path.abspath(__file__)), '..')
code = ("import sys, json; sys.path.insert(0, %r); from xanthos_amd import _hip; _hip.LIB_PATH = sys.argv[1]; sys.argv = ['bench.py', '--workload', 'pm_abcd', '--steps', '20', '--warmup', '3', '--no-cpu-baseline']; "
        "import runpy; runpy.run_path(%r, run_name='__main__')") % (root, os.path.join(root, 'bench.py'))
for rep in range(3):
    for lib in sys.argv[1:3]:
        out = subprocess.run([sys.executable, '-c', code, os.path.abspath(lib)], capture_output=True, text=True)
        try:
            d = json.loads(out.stdout.strip().splitlines()[-1])
            print(os.path.basename(lib), 'pm_pet %.4f ms  step %.4f ms' % (d['kernels']['pm_pet']['avg_ms'], d['ms_per_step']))
        except Exception as e:
            print(os.path.basename(lib), 'failed', out.stderr[-500:])
