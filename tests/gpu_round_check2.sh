# large-structure paths: configs[4] timings with and without the hand-over of k_deep in global memory, timelines
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/k2
python tests/gpu_config5.py > gpurun_out/k2/config5.log 2>&1; grep "configs\[4\]" gpurun_out/k2/config5.log
ENZYMM_AMD_DEEP_SHARE=0 python tests/gpu_config5.py > gpurun_out/k2/config5_noshare.log 2>&1; grep "configs\[4\]" gpurun_out/k2/config5_noshare.log
ENZYMM_AMD_DEEP_SHARE=2 python tests/gpu_config5.py > gpurun_out/k2/config5_share2.log 2>&1; grep "configs\[4\]" gpurun_out/k2/config5_share2.log
python tests/gpu_deep_timeline.py 1 > gpurun_out/k2/timeline1.log 2>&1; grep -v "Warning\|amdgpu.ids\|packed =" gpurun_out/k2/timeline1.log | tail -16
