# Instruction-fetch counters of the routing kernel, whole world vs one quarter shard (one unit per CU): run through gpurun.
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for sh in "" "0/4"; do
  export XH_STATS_SHARD=$sh
  [ -z "$sh" ] && unset XH_STATS_SHARD
  for set in "SQ_IFETCH SQ_IFETCH_LEVEL SQ_WAVE_CYCLES SQ_WAIT_INST_ANY" "SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_BUSY_CYCLES" "SQ_WAIT_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_INST_CYCLES_VMEM"; do
    rm -rf gpurun_out/prof_if
    timeout 600 rocprofv3 --pmc $set --output-format csv -d gpurun_out/prof_if -- python3 tools/flow_stats.py 240 > /dev/null 2>&1
    echo "== shard '${sh}' : $set"
    python3 tools/pmc_summary.py gpurun_out/prof_if | grep -E "k_mrtm_wave\("
  done
done
