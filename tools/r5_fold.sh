mkdir -p gpurun_out/r5fold
rm -f gpurun_out/r5fold/*
(XH_FLOW_FOLD=1 timeout 900 python -m pytest tests/test_gpu_reassoc.py -q -m gpu 2>&1 | tail -20) > gpurun_out/r5fold/tests.log
for rep in 1 2; do
  for fold in 0 1; do
    echo "fold=$fold $(XH_FLOW_FOLD=$fold timeout 300 python tools/rsum_probe.py 600 120 4 2>&1 | grep -E 'reassociated plan|reassoc  mrtm_route|PARITY|avg:|unit wall|SIMDs in use|cycles per sub-step outside' | tr '\n' ' ')" >> gpurun_out/r5fold/fold_ab.log
  done
done
(XH_FLOW_FOLD=1 timeout 600 python bench.py --steps 20 --warmup 3 2>&1 | tail -3) > gpurun_out/r5fold/bench.log
tail -15 gpurun_out/r5fold/tests.log; cat gpurun_out/r5fold/fold_ab.log; cut -c1-1500 gpurun_out/r5fold/bench.log
