# VALU wave-instructions of k_pm_pet per launch (config 2 bench), through gpurun
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rm -rf gpurun_out/prof_pm
timeout 600 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVES --output-format csv -d gpurun_out/prof_pm -- python3 bench.py --workload pm_abcd --steps 2 --warmup 1 --no-cpu-baseline > /dev/null 2>&1
python3 tools/pmc_summary.py gpurun_out/prof_pm | grep -E "k_pm_pet"
