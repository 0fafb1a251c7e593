cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout 600 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVES --output-format csv -d gpurun_out/prof_insts -- python3 bench.py --steps 2 --warmup 1 --order staged --no-cpu-baseline --no-end-to-end > /dev/null 2>&1
timeout 600 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY --output-format csv -d gpurun_out/prof_cyc -- python3 bench.py --steps 2 --warmup 1 --order staged --no-cpu-baseline --no-end-to-end > /dev/null 2>&1
python3 tools/pmc_summary.py gpurun_out/prof_insts | grep -E "k_pm_pet|k_abcd|k_mrtm_wave" 
python3 tools/pmc_summary.py gpurun_out/prof_cyc | grep -E "k_pm_pet|k_abcd|k_mrtm_wave"
