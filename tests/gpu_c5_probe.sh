# configs[4] probes: table-size census (diagnostic build) and kernel stats of ten complexes at the default cutoffs / one
# complex at the loose ones. Not a test.
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/c5
python tests/gpu_stats.py 200 > gpurun_out/c5/stats_headline200.log 2>&1
python tests/gpu_stats.py 10 complexes_default > gpurun_out/c5/stats_default.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/c5/prof -o c5 -- python3 tests/gpu_config5_only.py 10 default > gpurun_out/c5/run_default.log 2>&1
find gpurun_out/c5/prof -name "*kernel_stats.csv" -exec cp {} gpurun_out/c5/kernel_stats_default.csv \;
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/c5/prof2 -o c5 -- python3 tests/gpu_config5_only.py 1 > gpurun_out/c5/run_one_loose.log 2>&1
find gpurun_out/c5/prof2 -name "*kernel_stats.csv" -exec cp {} gpurun_out/c5/kernel_stats_one_loose.csv \;
rm -rf gpurun_out/c5/prof gpurun_out/c5/prof2
grep -v "^W2026" gpurun_out/c5/run_default.log | tail -5
