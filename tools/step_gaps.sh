cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 600 rocprofv3 --kernel-trace --output-format csv -d gpurun_out/gap_fed -- python3 bench.py --steps 10 --warmup 6 --no-cpu-baseline --no-end-to-end > gpurun_out/gap_fed.json 2> gpurun_out/gap_fed.log
timeout 600 rocprofv3 --kernel-trace --output-format csv -d gpurun_out/gap_staged -- python3 bench.py --steps 10 --warmup 6 --order staged --no-cpu-baseline --no-end-to-end > gpurun_out/gap_staged.json 2> gpurun_out/gap_staged.log
python3 tools/step_gaps.py gpurun_out/gap_fed 9 > gpurun_out/gaps_fed.txt 2>&1
python3 tools/step_gaps.py gpurun_out/gap_staged > gpurun_out/gaps_staged.txt 2>&1
head -3 gpurun_out/gap_fed/*/*kernel_trace.csv | cut -c1-300
cat gpurun_out/gaps_fed.txt gpurun_out/gaps_staged.txt
