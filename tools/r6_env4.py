import os, subprocess, sys, json
sys.path.insert(0, '/root/repo/tests')
import test_gpu_reassoc as t
open('/tmp/child4.py', 'w').write(t._REASSOC_PARTITION_CHILD)
for rep in range(3):
    for env in ({'XH_FLOW_RS': '2048', 'XH_FLOW_SPARE': '3'}, {'XH_FLOW_RS': '2048'}, {'XH_FLOW_SPARE': '3'}):
        e = dict(os.environ, XH_FLOW_DEBUG='1'); e.update(env)
        import time; t0 = time.time()
        out = subprocess.run([sys.executable, '/tmp/child4.py', '/root/repo'], env=e, capture_output=True, text=True, timeout=300)
        print(env, 'rc', out.returncode, '%.1f s' % (time.time() - t0), out.stdout.strip().splitlines()[-1:] )
        err = [l for l in out.stderr.splitlines() if 'fault' in l.lower() or 'not used' in l or 'rerout' in l.lower() or 'pair unit' in l or 'single-sum' in l]
        print('\n'.join(err[-8:]))
