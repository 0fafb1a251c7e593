# Round-5 artefacts (run through gpurun from the repo root): the GPU suite, the round's profile set (tools/profile_round.sh),
# one shard of N of the world on this GPU, the floors.
mkdir -p gpurun_out/r5f
(timeout 2400 python -m pytest tests -q -m gpu 2>&1 | tail -12) > gpurun_out/r5f/tests.log
bash tools/profile_round.sh > gpurun_out/r5f/profile_round.log 2>&1
for sh in 0/2 0/4 0/8; do
  echo "shard $sh: $(XH_STATS_SHARD=$sh XH_STATS_ROUTE_SPIN=120 XH_STATS_ABCD_SPIN=120 timeout 300 python3 tools/flow_stats.py 600 2>&1 | grep -E '^route ms|^shard')" >> gpurun_out/r5f/shards.txt
done
./tools/micro/substep_rsum.bin > gpurun_out/r5f/substep_rsum.txt 2>&1
timeout 600 python3 tools/run_model_bench.py > gpurun_out/r5f/run_model_bench.txt 2>&1
tail -5 gpurun_out/r5f/tests.log; cat gpurun_out/r5f/shards.txt; tail -12 gpurun_out/r5f/run_model_bench.txt; head -c 700 gpurun_out/bench.json; echo; tail -3 gpurun_out/bench.log
