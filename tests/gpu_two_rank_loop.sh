# Debug aid: the two-rank bench child of test_bench_two_ranks_as_the_driver_launches_it, several times in a row, each under
# its own timeout, with the library's phase trace on stderr.  Not a test.
#   bash tests/gpu_two_rank_loop.sh [runs] [seconds per run]
runs=${1:-6}; limit=${2:-100}
for i in $(seq $runs); do
  port=$((29700 + i))
  start=$(date +%s)
  ENZYMM_AMD_BENCH_ONE_GPU=1 ENZYMM_AMD_TRACE=1 HSA_ENABLE_IPC_MODE_LEGACY=0 timeout -s KILL $limit python -m torch.distributed.run --nnodes=1 --nproc-per-node=2 --master-addr 127.0.0.1 --master-port $port bench.py --gpus 2 --structures 64 --steps 2 --warmup 1 --no-cpu-baseline --no-extras --streamed-passes 0 --sustained-seconds 0 > gpurun_out/two_$i.out 2> gpurun_out/two_$i.err
  rc=$?
  echo "run $i: rc $rc after $(( $(date +%s) - start )) s, $(grep -c '^{' gpurun_out/two_$i.out) JSON line(s)"
  if [ $rc -ne 0 ]; then tail -n 12 gpurun_out/two_$i.err | cut -c1-220; fi
done
