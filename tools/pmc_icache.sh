cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rocprofv3 --list-avail 2>/dev/null | grep -i -E "ICACHE|IFETCH|SQC_" | head -60 > gpurun_out/avail_ic.txt
timeout 600 rocprofv3 --pmc SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQ_IFETCH --output-format csv -d gpurun_out/prof_ic -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline > /dev/null 2>&1
python3 tools/pmc_summary.py gpurun_out/prof_ic | grep -E "k_pm_pet|k_abcd|k_mrtm_skew"
timeout 600 rocprofv3 --pmc SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_INSTS_VALU SQ_WAVE_CYCLES --output-format csv -d gpurun_out/prof_ic2 -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline > /dev/null 2>&1
python3 tools/pmc_summary.py gpurun_out/prof_ic2 | grep -E "k_mrtm_skew"
