mkdir -p gpurun_out/r5e
timeout 300 python tools/r5_dbg.py 2>&1 | tail -5 > gpurun_out/r5e/dbg.log
(timeout 900 python -m pytest tests/test_gpu_multirank.py tests/test_gpu_parity.py -q -m gpu -k "n_ranks or cross_checked or self_loop" 2>&1 | tail -25) > gpurun_out/r5e/tests_new.log
rm -f gpurun_out/r5e/fence_ab.log
for rep in 1 2 3; do
  for mode in default mid 1; do
    if [ $mode = default ]; then unset XH_ROUTE_FENCED; else export XH_ROUTE_FENCED=$mode; fi
    echo "fence=$mode $(timeout 300 python tools/rsum_probe.py 600 120 4 2>&1 | grep -E 'reassoc  mrtm_route|^exact    mrtm_route|PARITY' | tr '\n' ' ')" >> gpurun_out/r5e/fence_ab.log
  done
done
unset XH_ROUTE_FENCED
for lib in libxanthos_hip.so libxanthos_hip_ch128.so libxanthos_hip_ch512.so libxanthos_hip.so libxanthos_hip_ch128.so libxanthos_hip_ch512.so; do
  echo "$lib $(XH_LIBRARY=$PWD/xanthos_amd/$lib timeout 300 python tools/rsum_probe.py 600 120 4 2>&1 | grep -E 'reassoc  mrtm_route|PARITY' | tr '\n' ' ')" >> gpurun_out/r5e/ch_ab.log
done
(timeout 1800 python -m pytest tests -q -m gpu 2>&1 | tail -15) > gpurun_out/r5e/tests_all.log
cat gpurun_out/r5e/dbg.log; tail -8 gpurun_out/r5e/tests_new.log; cat gpurun_out/r5e/fence_ab.log gpurun_out/r5e/ch_ab.log; tail -6 gpurun_out/r5e/tests_all.log
