# Round-6 artefacts (run through gpurun from the repo root): the GPU suite, the round's profile set (tools/profile_round.sh),
# the per-unit cycle table of the single-sum plan and its A/Bs, one shard of N of the world on this GPU,
# the lone-wave floors, the differential fuzz of the default form.
mkdir -p gpurun_out/r6f
export XH_CACHE_DIR=/tmp/xh_cache_r6
(timeout 2400 python -m pytest tests -q -m gpu -p no:cacheprovider 2>&1 | tail -12) > gpurun_out/r6f/tests.log
bash tools/profile_round.sh > gpurun_out/r6f/profile_round.log 2>&1
./tools/micro/substep_rsum.bin > gpurun_out/r6f/substep_rsum.txt 2>&1
# the plan the library ships against its own switches, alternating on this box (tools/rsum_probe.py: parity against the bit-exact
# kernel of the same library, mrtm_route alone, per-unit cycles): single sums with a CU per pair unit / without / the plan of pairs
for rep in 1 2; do
  for v in "XH_RSUM_EXCL=1" "XH_RSUM_EXCL=0" "XH_RSUM_SINGLE=0"; do
    echo "== $v" >> gpurun_out/r6f/single_sum_ab.txt
    env $v XH_FLOW_CHECK=1 timeout 600 python3 tools/rsum_probe.py 600 120 3 2>&1 | grep -v "^  LDS ops" >> gpurun_out/r6f/single_sum_ab.txt
  done
done
XH_FLOW_DEBUG=1 timeout 600 python3 tools/rsum_probe.py 600 120 3 > gpurun_out/r6f/unit_cycles.txt 2>&1
for sh in 0/2 0/4 0/8; do
  echo "shard $sh: $(XH_STATS_SHARD=$sh XH_STATS_ROUTE_SPIN=120 XH_STATS_ABCD_SPIN=120 timeout 300 python3 tools/flow_stats.py 600 2>&1 | grep -E '^route ms|^shard')" >> gpurun_out/r6f/shards.txt
done
timeout 1500 python3 tools/fuzz_reassoc.py 210 4242 > gpurun_out/r6f/fuzz_reassoc.txt 2>&1
timeout 900 python3 tools/fuzz_routing.py 60 > gpurun_out/r6f/fuzz_exact.txt 2>&1
timeout 600 python3 tools/run_model_bench.py > gpurun_out/r6f/run_model_bench.txt 2>&1
tail -5 gpurun_out/r6f/tests.log; cat gpurun_out/r6f/shards.txt; tail -3 gpurun_out/r6f/fuzz_reassoc.txt; head -c 700 gpurun_out/bench.json; echo; tail -3 gpurun_out/bench.log
