mkdir -p gpurun_out/r5b
(timeout 1500 python -m pytest tests -x -q -m gpu 2>&1 | tail -15) > gpurun_out/r5b/tests.log
./tools/micro/substep_rsum.bin > gpurun_out/r5b/micro.log 2>&1
(timeout 300 python tools/rsum_probe.py 600 120 3 2>&1 | tail -32) > gpurun_out/r5b/probe_default.log
(XH_WAVE_PRIO=0 timeout 300 python tools/rsum_probe.py 600 120 3 2>&1 | tail -32) > gpurun_out/r5b/probe_noprio.log
for cap in 64 36 32 24; do (XH_FLOW_PIECE_CAP=$cap timeout 300 python tools/rsum_probe.py 600 120 3 2>&1 | tail -32) > gpurun_out/r5b/probe_cap$cap.log; done
(timeout 600 python tools/pm_ab.py xanthos_amd/libxanthos_hip.so xanthos_amd/libxanthos_hip_contract.so 2>&1 | tail -8) > gpurun_out/r5b/pm_ab.log
tail -5 gpurun_out/r5b/tests.log; cat gpurun_out/r5b/micro.log; grep -h "mrtm_route ms\|PARITY\|shared SIMDs" gpurun_out/r5b/probe_*.log; cat gpurun_out/r5b/pm_ab.log
