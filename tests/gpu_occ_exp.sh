# A/B of the search kernels' occupancy on small structures (tests/gpu_occupancy.py).  Not a test.
for mean in 100 150; do
  echo "=== baseline 16 waves, 160 KB, mean $mean"; python tests/gpu_occupancy.py 1000 $mean 2>&1 | tail -4
  echo "=== occ 12 waves x2 (80 KB), mean $mean"; ENZYMM_AMD_LIB=$PWD/enzymm_amd/libenzymm_amd_occ.so ENZYMM_AMD_LDS_BUDGET=81920 python tests/gpu_occupancy.py 1000 $mean 2>&1 | tail -4
  echo "=== occ 12 waves x1 (160 KB), mean $mean"; ENZYMM_AMD_LIB=$PWD/enzymm_amd/libenzymm_amd_occ.so python tests/gpu_occupancy.py 1000 $mean 2>&1 | tail -4
done
