# A/B of ENZYMM_AMD_READ_BOUNCE (files read through a cache-resident buffer of that many KB, streaming stores into the
# page-locked buffer; 0 = read() straight into it): run_files over 16 000 files, alternating.  Not a test.
cd $GRAFT_REPO_ROOT
for round in 1 2; do
  for b in ${@:-0 96}; do
    echo "READ_BOUNCE=$b KB"
    ENZYMM_AMD_READ_BOUNCE=$b python tests/gpu_end_to_end.py 16000 0 2>&1 | grep "^run_files"
  done
done
