import sys, json
d=json.loads(sys.stdin.read())
print(d["ms_per_step"], d["roofline"]["us_per_substep"], {k:round(v["avg_ms"],3) for k,v in d["kernels"].items()})
