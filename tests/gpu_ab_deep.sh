# A/B of the depth-first part: lane = triple (default) against the wave-cooperative search (ENZYMM_AMD_DEEP=wave).  Not a test.
echo "=== lanes"; python tests/gpu_occupancy.py 1000 300 2>&1 | tail -4
echo "=== wave"; ENZYMM_AMD_DEEP=wave python tests/gpu_occupancy.py 1000 300 2>&1 | tail -4
