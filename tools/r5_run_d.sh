mkdir -p gpurun_out/r5d
(timeout 900 python -m pytest tests/test_gpu_multirank.py tests/test_gpu_parity.py -q -m gpu -k "n_ranks or self_loop or comm or side_gather or fault_inside or fed_routing_equals or pm_ or bench_gpus" 2>&1 | tail -25) > gpurun_out/r5d/tests_new.log
(timeout 1800 python -m pytest tests -q -m gpu -x 2>&1 | tail -15) > gpurun_out/r5d/tests_all.log
(timeout 600 python tools/pm_ab.py xanthos_amd/libxanthos_hip_pmnc.so xanthos_amd/libxanthos_hip.so 2>&1 | tail -8) > gpurun_out/r5d/pm_ab.log
tail -12 gpurun_out/r5d/tests_new.log; tail -6 gpurun_out/r5d/tests_all.log; cat gpurun_out/r5d/pm_ab.log
