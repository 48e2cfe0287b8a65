#!/bin/bash
# on the GPU box: bash tools/probe/sdf_variants.sh  -> gpurun_out/sdf_variants.txt
cd ${GRAFT_REPO_ROOT:-.}
mkdir -p /tmp/sv gpurun_out
C="/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -std=c++17 -fPIC -shared -I d3human-code_amd/csrc d3human-code_amd/csrc/sdf_mlp.hip d3human-code_amd/csrc/timing.hip"
$C -o /tmp/sv/base.so &
$C -DD3H_SDF_GLDS=0 -o /tmp/sv/regstage.so &
$C -DD3H_PROBE_NO_EPI -o /tmp/sv/noepi.so &
$C -DD3H_PROBE_NO_STAGE -o /tmp/sv/nostage.so &
$C -DD3H_PROBE_NO_STAGE -DD3H_PROBE_NO_BARRIER -o /tmp/sv/nostage_nobar.so &
$C -DD3H_PROBE_NO_STAGE -DD3H_PROBE_NO_BARRIER -DD3H_PROBE_NO_EPI -o /tmp/sv/mfma_only.so &
wait
/opt/rocm/bin/hipcc -O2 -o /tmp/sv/probe tools/probe/sdf_variants.cpp -ldl
for v in base regstage noepi nostage nostage_nobar mfma_only; do
  cd /tmp/sv && ./probe ./$v.so 262144; ./probe ./$v.so 50000 | tail -1; cd - > /dev/null
done > gpurun_out/sdf_variants.txt 2>&1
