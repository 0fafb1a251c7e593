#!/bin/bash
# Collect the round's profiling artefacts on the GPU box (run through gpurun from the repo root):
#   kernel trace + stats of the default bench command, then FETCH_SIZE / WRITE_SIZE in two separate PMC passes.
# Outputs land in gpurun_out/prof_* ; tools/pmc_to_json.py and a copy of *_kernel_stats.csv go to profiles/<round>/.
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cd $R
mkdir -p gpurun_out
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_trace -- python3 bench.py --steps 5 --warmup 1 --no-cpu-baseline > gpurun_out/bench_under_rocprof.json 2> gpurun_out/bench_under_rocprof.log
timeout 600 rocprofv3 --pmc FETCH_SIZE --output-format csv -d gpurun_out/prof_fetch -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline > /dev/null 2>&1
timeout 600 rocprofv3 --pmc WRITE_SIZE --output-format csv -d gpurun_out/prof_write -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline > /dev/null 2>&1
python3 tools/pmc_to_json.py gpurun_out/prof_fetch gpurun_out/prof_write gpurun_out/pmc_traffic.json > /dev/null
cp gpurun_out/prof_trace/*/*kernel_stats.csv gpurun_out/kernel_stats.csv
timeout 300 python3 bench.py --steps 5 --warmup 1 > gpurun_out/bench.json 2> gpurun_out/bench.log
XH_FLOW_STATS=1 timeout 300 python3 tools/flow_stats.py 120 > gpurun_out/flow_unit_cycles.txt 2>&1
timeout 300 python3 tools/bench_calib.py > gpurun_out/calib_config5.txt 2>&1
head -c 600 gpurun_out/bench.json; echo; head -8 gpurun_out/kernel_stats.csv | cut -c1-160; cat gpurun_out/pmc_traffic.json | head -50
