#!/bin/bash
# Collect the round's profiling artefacts on the GPU box (run through gpurun from the repo root):
#   kernel trace + stats of the default bench command, FETCH_SIZE / WRITE_SIZE and the instruction counters in separate
#   PMC passes (never together with a trace), the per-unit routing cycles, and the calibration bench line.
# Outputs land in gpurun_out/ ; copy what is to be judged into profiles/<round>/ (README there).
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cd $R
mkdir -p gpurun_out
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_trace -- python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-end-to-end --no-secondary > gpurun_out/bench_under_rocprof.json 2> gpurun_out/bench_under_rocprof.log
# (counter passes serialise the kernels of a process: the fed order's routing kernel would wait for months that cannot be
#  produced beside it -- the passes run the stages one after the other, which is also what per-kernel counters are about)
timeout 600 rocprofv3 --pmc FETCH_SIZE --output-format csv -d gpurun_out/prof_fetch -- python3 bench.py --steps 2 --warmup 1 --order staged --no-cpu-baseline --no-end-to-end --no-secondary > /dev/null 2>&1
timeout 600 rocprofv3 --pmc WRITE_SIZE --output-format csv -d gpurun_out/prof_write -- python3 bench.py --steps 2 --warmup 1 --order staged --no-cpu-baseline --no-end-to-end --no-secondary > /dev/null 2>&1
# counter calibration on the streams' own access shapes: 1 GiB (past the Infinity Cache) and 128 MiB
for sz in 1024 128; do
  timeout 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d gpurun_out/cal_fetch_$sz -- tools/micro/sc1_traffic.bin $sz > gpurun_out/cal_bytes_$sz.txt 2>&1
  timeout 300 rocprofv3 --pmc WRITE_SIZE --output-format csv -d gpurun_out/cal_write_$sz -- tools/micro/sc1_traffic.bin $sz > /dev/null 2>&1
done
BYTES=$(grep bytes_per_launch gpurun_out/cal_bytes_1024.txt | cut -d" " -f2)
python3 tools/pmc_to_json.py gpurun_out/prof_fetch gpurun_out/prof_write gpurun_out/pmc_traffic.json gpurun_out/cal_fetch_1024 gpurun_out/cal_write_1024 $BYTES > /dev/null
BYTES2=$(grep bytes_per_launch gpurun_out/cal_bytes_128.txt | cut -d" " -f2)
python3 tools/pmc_to_json.py gpurun_out/prof_fetch gpurun_out/prof_write gpurun_out/pmc_traffic_cal128.json gpurun_out/cal_fetch_128 gpurun_out/cal_write_128 $BYTES2 > /dev/null
cp gpurun_out/prof_trace/*/*kernel_stats.csv gpurun_out/kernel_stats.csv
timeout 600 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVES --output-format csv -d gpurun_out/prof_insts -- python3 bench.py --steps 2 --warmup 1 --order staged --no-cpu-baseline --no-end-to-end --no-secondary > /dev/null 2>&1
timeout 600 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY --output-format csv -d gpurun_out/prof_cyc -- python3 bench.py --steps 2 --warmup 1 --order staged --no-cpu-baseline --no-end-to-end --no-secondary > /dev/null 2>&1
# the calibration marches (config 5): instruction counts for the measured issue fraction of bench.py's `calib` lines
timeout 600 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVES --output-format csv -d gpurun_out/prof_insts_calib -- python3 bench.py --workload calib --steps 2 --warmup 1 --no-cpu-baseline > /dev/null 2>&1
timeout 600 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY --output-format csv -d gpurun_out/prof_cyc_calib -- python3 bench.py --workload calib --steps 2 --warmup 1 --no-cpu-baseline > /dev/null 2>&1
python3 tools/pmc_insts_json.py gpurun_out/pmc_insts.json gpurun_out/prof_insts gpurun_out/prof_cyc gpurun_out/prof_insts_calib gpurun_out/prof_cyc_calib > /dev/null
(python3 tools/pmc_summary.py gpurun_out/prof_insts; python3 tools/pmc_summary.py gpurun_out/prof_cyc) | grep -E "k_pm_pet|k_abcd|k_mrtm_wave|k_mrtm_rsum|k_mrtm_units" > gpurun_out/pmc_insts.txt
(python3 tools/pmc_summary.py gpurun_out/prof_insts_calib; python3 tools/pmc_summary.py gpurun_out/prof_cyc_calib) | grep -E "k_calib" >> gpurun_out/pmc_insts.txt
timeout 900 python3 bench.py --steps 20 --warmup 5 > gpurun_out/bench.json 2> gpurun_out/bench.log
timeout 300 python3 bench.py --workload pm_abcd --steps 10 --warmup 2 > gpurun_out/bench_pm_abcd.json 2> gpurun_out/bench_pm_abcd.log
timeout 600 python3 bench.py --workload calib --steps 5 --warmup 1 > gpurun_out/bench_calib.json 2> gpurun_out/bench_calib.log
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_trace_calib -- python3 bench.py --workload calib --steps 5 --warmup 1 --no-cpu-baseline > /dev/null 2>&1
cp gpurun_out/prof_trace_calib/*/*kernel_stats.csv gpurun_out/kernel_stats_calib.csv
timeout 300 python3 bench.py --steps 20 --warmup 5 --order staged --no-end-to-end --no-cpu-baseline > gpurun_out/bench_staged.json 2> /dev/null
timeout 300 python3 tools/feed_interference.py 2>&1 | grep -E "stage by stage|fed" > gpurun_out/feed_interference.txt
timeout 600 python3 tools/feed_stress.py 150 2>&1 | tail -3 > gpurun_out/feed_stress.txt
XH_FLOW_STATS=1 timeout 300 python3 tools/flow_stats.py 720 > gpurun_out/flow_unit_cycles.txt 2>&1
XH_STATS_ROUTE_SPIN=120 XH_STATS_ABCD_SPIN=120 XH_STATS_LOOP=5 timeout 300 python3 tools/flow_stats.py 600 2>&1 | grep -E "back-to-back|histogram|wave\(s\)" > gpurun_out/flow_pipelined.txt
head -c 600 gpurun_out/bench.json; echo; head -8 gpurun_out/kernel_stats.csv | cut -c1-160; cat gpurun_out/pmc_traffic.json | head -50
# the N = 2 dry run on this box's one GPU (both ranks on device 0, gloo process group, RCCL stand-in named by path)
make -C tests/fake_rccl > /dev/null 2>&1
XH_BENCH_ONE_DEVICE=1 XH_BENCH_BACKEND=gloo XH_RCCL_LIBRARY=$R/tests/fake_rccl/librccl.so.1 XH_FAKE_RCCL_DIR=/tmp timeout 900 python3 bench.py --gpus 2 --steps 3 --warmup 1 --route-flags 4 --check-gather --no-replica-figure > gpurun_out/bench_2rank_dryrun.json 2> gpurun_out/bench_2rank_dryrun.log
head -c 400 gpurun_out/bench_2rank_dryrun.json; echo
