mkdir -p gpurun_out/r6
timeout 600 python tools/dbg/gpu_ssim_occ_probe.py > gpurun_out/r6/ssim_occ_probe.txt 2>&1
cd /tmp && export TMPDIR=/tmp
D3H_SSIM_OCC=1 timeout 600 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/r6/prof_occ1 -o occ1 -- python3 $GRAFT_REPO_ROOT/tools/dbg/gpu_ssim_occ_probe.py > /dev/null 2>&1
cd $GRAFT_REPO_ROOT
find gpurun_out/r6/prof_occ1 -name "*kernel_stats*" | head -1 | xargs -I{} sh -c 'head -12 {} | cut -c1-200' >> gpurun_out/r6/ssim_occ_probe.txt
find gpurun_out/r6/prof_occ1 -type f ! -name "*kernel_stats*" -delete
