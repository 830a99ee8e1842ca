# A/B of two builds of the library on one box: bench.py's timed region only, alternating.  Not a test.
#   bash tests/gpu_ab_bench.sh <variant name> [rounds]      (libenzymm_amd_<variant>.so against libenzymm_amd.so)
v=$1; rounds=${2:-3}
for r in $(seq $rounds); do
  for lib in "" "_$v"; do
    for p in 2 1; do
      ENZYMM_AMD_LIB=$PWD/enzymm_amd/libenzymm_amd$lib.so python bench.py --steps 20 --warmup 5 --pipeline $p --no-extras --no-cpu-baseline --sustained-seconds 0 --streamed-passes 0 2>/dev/null | grep "^{" | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('lib%-8s pipeline $p: %.0f structures/s, %.3f ms per step' % ('$lib', d['value'], d['ms_per_step']))"
    done
  done
done
