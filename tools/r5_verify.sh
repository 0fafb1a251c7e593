# final verification of the round's last build: the GPU suite and the driver's bench command
mkdir -p gpurun_out/r5v
(timeout 2400 python -m pytest tests -q -m gpu 2>&1 | tail -6) > gpurun_out/r5v/tests.log
timeout 900 python3 bench.py --steps 20 --warmup 5 > gpurun_out/r5v/bench.json 2> gpurun_out/r5v/bench.log; echo "rc $?" >> gpurun_out/r5v/bench.log
python3 -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/r5v/smoke.log 2>&1
tail -3 gpurun_out/r5v/tests.log; tail -2 gpurun_out/r5v/bench.log | cut -c1-300; head -c 400 gpurun_out/r5v/bench.json; echo; tail -2 gpurun_out/r5v/smoke.log
