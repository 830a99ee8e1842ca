cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/k1
timeout 1500 python -m pytest tests -m gpu -x -q > gpurun_out/k1/tests.log 2>&1; echo "tests rc=$?"
tail -3 gpurun_out/k1/tests.log
python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extras --streamed-passes 0 --sustained-seconds 0 > gpurun_out/k1/bench.json 2> gpurun_out/k1/bench.err; tail -c 600 gpurun_out/k1/bench.json
python tests/gpu_config5.py > gpurun_out/k1/config5.log 2>&1; grep "configs\[4\]" gpurun_out/k1/config5.log
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/k1/prof -o b -- python3 bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-extras --streamed-passes 0 --sustained-seconds 0 --pipeline 1 > gpurun_out/k1/trace.log 2>&1
find gpurun_out/k1/prof -name "*kernel_stats.csv" -exec cp {} gpurun_out/k1/kernel_stats_p1.csv \;
rm -rf gpurun_out/k1/prof
