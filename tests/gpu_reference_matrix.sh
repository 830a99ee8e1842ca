# The reference's call pattern under a few settings (tests/gpu_reference_pattern.py).  Not a test.
for grain in 65536 8192 4096; do
  echo "=== host grain $grain"; ENZYMM_AMD_HOST_GRAIN=$grain python tests/gpu_reference_pattern.py 256 16 2>&1 | grep -E "molecules x|of a launch"
done
