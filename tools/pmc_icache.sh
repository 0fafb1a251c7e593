cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout 600 rocprofv3 --pmc SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQ_WAIT_INST_ANY --output-format csv -d gpurun_out/prof_ic -- python3 bench.py --steps 2 --warmup 1 --order staged --no-cpu-baseline --no-end-to-end > /dev/null 2>&1
python3 tools/pmc_summary.py gpurun_out/prof_ic | grep -E "k_mrtm_wave\("
