mkdir -p gpurun_out/r5e
(timeout 900 python -m pytest tests/test_gpu_multirank.py tests/test_gpu_parity.py -q -m gpu -k "n_ranks or cross_checked or self_loop" 2>&1 | tail -25) > gpurun_out/r5e/tests_new.log
# fence A/B on the reassociated kernel: default (vmcnt(8) + publication lag) against vmcnt(0) in front of every publication,
# and the full release / acquire form, same box, alternating
for rep in 1 2 3; do
  for mode in default mid 1; do
    if [ $mode = default ]; then unset XH_ROUTE_FENCED; else export XH_ROUTE_FENCED=$mode; fi
    echo "fence=$mode $(timeout 300 python tools/rsum_probe.py 600 120 4 2>&1 | grep -E 'reassoc  mrtm_route|^exact    mrtm_route|PARITY')" >> gpurun_out/r5e/fence_ab.log
  done
done
unset XH_ROUTE_FENCED
(timeout 1800 python -m pytest tests -q -m gpu 2>&1 | tail -15) > gpurun_out/r5e/tests_all.log
tail -8 gpurun_out/r5e/tests_new.log; cat gpurun_out/r5e/fence_ab.log; tail -6 gpurun_out/r5e/tests_all.log
