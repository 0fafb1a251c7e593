#!/bin/bash
# A/B of the lane optimiser: mrtm_route alone (staged order), alternating
mkdir -p gpurun_out
for rep in 1 2; do
for T in 0 1000; do
  XH_FLOW_LANE_TRIALS=$T timeout 300 python bench.py --steps 20 --warmup 6 --no-end-to-end --no-cpu-baseline --order staged > gpurun_out/ab_lane_${T}_${rep}.json 2> gpurun_out/ab_lane_${T}_${rep}.err
  python tools/bench_brief.py < gpurun_out/ab_lane_${T}_${rep}.json | sed "s/^/T=$T rep=$rep: /"
done
done
