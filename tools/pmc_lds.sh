cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rocprofv3 --list-avail 2>/dev/null | grep -E "Counter_Name" | grep -i -E "LDS" > gpurun_out/avail_lds.txt
timeout 600 rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_ACTIVE_INST_LDS --output-format csv -d gpurun_out/prof_lds -- python3 bench.py --steps 2 --warmup 1 --order staged --no-cpu-baseline --no-end-to-end > /dev/null 2>&1
python3 tools/pmc_summary.py gpurun_out/prof_lds | grep -E "k_mrtm_wave"
timeout 600 rocprofv3 --pmc SQ_WAIT_INST_LDS SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL SQ_WAVE_CYCLES --output-format csv -d gpurun_out/prof_lds2 -- python3 bench.py --steps 2 --warmup 1 --order staged --no-cpu-baseline --no-end-to-end > /dev/null 2>&1
python3 tools/pmc_summary.py gpurun_out/prof_lds2 | grep -E "k_mrtm_wave"
