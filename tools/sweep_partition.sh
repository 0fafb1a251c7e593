cd $GRAFT_REPO_ROOT
export XH_ROUTE_VALIDATE_FIRST=0 XH_ROUTE_LEARN_CACHE=0
run() { timeout 120 python3 bench.py --steps 10 --warmup 6 --order staged --no-end-to-end --no-cpu-baseline 2>/dev/null | python3 -c "
import json,sys
try:
    d=json.loads(sys.stdin.read()); print('$1', round(d['kernels']['mrtm_route']['avg_ms'],3), d['routing_plan']['flow_units'], d['routing_plan']['flow_edges'], d['routing_plan']['form'][:28], d['routing_plan']['guard_trips'])
except Exception as e: print('$1', 'failed', e)"; }
run "default"
for mr in 4 6; do XH_FLOW_PLAIN_MIN_READS=$mr run "min_reads=$mr"; done
for tl in 4 6; do XH_FLOW_TLIMIT=$tl run "tlimit=$tl"; done
for tp in 5 7; do XH_FLOW_TLIMIT_PLAIN=$tp run "tlimit_plain=$tp"; done
for pc in 32 40 48 64; do XH_FLOW_PIECE_CAP=$pc run "piece_cap=$pc"; done
XH_FLOW_CUTRULE=0 run "cutrule=0"
XH_FLOW_CHAIN=0 run "chain=0"
XH_FLOW_RS=16384 run "rs=16384"
XH_FLOW_SPARE=64 run "spare=64"
run "default again"
