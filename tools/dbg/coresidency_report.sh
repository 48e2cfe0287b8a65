#!/bin/bash
# The co-residency hazard of the bf16-MFMA weight-gradient kernel, as counts: ticks (of 96 per line) whose gradients differ
# from the first tick of the same scene state, for the default build and for builds in which the kernel shares its SIMDs with other kernels.
#   bash tools/build_variant.sh share -DD3H_DWX_SHARE_SIMDS; ... share_nowork / share_noflush likewise;  gpurun -- 'bash tools/dbg/coresidency_report.sh'
for v in hip share share_nowork share_noflush; do
  for i in 1 2; do
    n=$(D3H_LIB_PATH=$PWD/d3human-code_amd/d3h/libd3h_$v.so python tools/dbg/gpu_dbg_x3_race.py 48 2>&1 | grep -c "bad: \[(")
    echo "build $v run $i: $n of 96 ticks differ"
  done
done
echo "exact-f32 weight-gradient kernel in its place (D3H_DW_X3=0), sharing build:"
for i in 1 2; do
  n=$(D3H_DW_X3=0 D3H_LIB_PATH=$PWD/d3human-code_amd/d3h/libd3h_share.so python tools/dbg/gpu_dbg_x3_race.py 48 2>&1 | grep -c "bad: \[(")
  echo "build share, D3H_DW_X3=0, run $i: $n of 96 ticks differ"
done
echo "streams serialised (D3H_NO_SIDE_STREAM=1), sharing build:"
n=$(D3H_NO_SIDE_STREAM=1 D3H_LIB_PATH=$PWD/d3human-code_amd/d3h/libd3h_share.so python tools/dbg/gpu_dbg_x3_race.py 48 2>&1 | grep -c "bad: \[(")
echo "build share, D3H_NO_SIDE_STREAM=1: $n of 96 ticks differ"
