mkdir -p gpurun_out/r5c
(timeout 2400 python -m pytest tests -q -m gpu 2>&1 | tail -30) > gpurun_out/r5c/tests.log
timeout 900 python3 bench.py --steps 20 --warmup 5 > gpurun_out/r5c/bench.json 2> gpurun_out/r5c/bench.log; echo "bench rc $?" >> gpurun_out/r5c/bench.log
(XH_LIBRARY=$PWD/xanthos_amd/libxanthos_hip_contract.so timeout 1500 python -m pytest tests/test_gpu_parity.py tests/test_gpu_end_to_end.py tests/test_gpu_calib.py tests/test_gpu_post.py -q -m gpu 2>&1 | tail -15) > gpurun_out/r5c/tests_contract.log
(XH_LIBRARY=$PWD/xanthos_amd/libxanthos_hip_contract.so timeout 600 python -m pytest tests/test_gpu_fullsize.py -q -m gpu -k "pm_abcd_parity or fused_pipeline or eight_shards" 2>&1 | tail -8) >> gpurun_out/r5c/tests_contract.log
tail -6 gpurun_out/r5c/tests.log; tail -4 gpurun_out/r5c/bench.log; head -c 1500 gpurun_out/r5c/bench.json; echo; tail -12 gpurun_out/r5c/tests_contract.log
