# one call after a kernel change: the GPU suite, a bench line, configs[4], kernel stats of a step with one search in flight
cd $GRAFT_REPO_ROOT
out=gpurun_out/${1:-k1}
mkdir -p $out
timeout 1500 python -m pytest tests -m gpu -x -q > $out/tests.log 2>&1; echo "tests rc=$?"
tail -3 $out/tests.log
python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extras --streamed-passes 0 --sustained-seconds 0 > $out/bench.json 2> $out/bench.err; python -c "
import json,sys
d=json.loads(open('$out/bench.json').read().strip().splitlines()[-1]); print('bench', round(d['value']), d['ms_per_step'])"
python tests/gpu_config5.py > $out/config5.log 2>&1; grep "configs\[4\]" $out/config5.log
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --stats --output-format csv -d $out/prof -o b -- python3 bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-extras --streamed-passes 0 --sustained-seconds 0 --pipeline 1 > $out/trace.log 2>&1
find $out/prof -name "*kernel_stats.csv" -exec cp {} $out/kernel_stats_p1.csv \;
rm -rf $out/prof
python - <<PY
import csv
for r in csv.DictReader(open("$out/kernel_stats_p1.csv")):
    if float(r["AverageNs"]) > 15000: print("  %-44s calls %4s avg %9.1f us" % (r["Name"][:44], r["Calls"], float(r["AverageNs"])/1e3))
PY
