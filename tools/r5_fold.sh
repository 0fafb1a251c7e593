mkdir -p gpurun_out/r5single
rm -f gpurun_out/r5single/*
for rep in 1 2; do
  for hv in 0 1; do
    echo "heavy=$hv $(XH_ROUTE_VALIDATE_FIRST=0 XH_EXP_HEAVY=$hv timeout 300 python tools/rsum_probe.py 600 120 4 2>&1 | grep -E 'reassociated plan|reassoc  mrtm_route|unit wall|cycles per sub-step outside' | tr '\n' ' ')" >> gpurun_out/r5single/heavy.log
  done
done
cat gpurun_out/r5single/heavy.log
