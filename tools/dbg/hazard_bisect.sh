#!/bin/bash
# reproducer v2 (diagnostic victims) against the sharing build and its probe variants;  gpurun -- 'bash tools/dbg/hazard_bisect.sh'
cd ${GRAFT_REPO_ROOT:-.}
O=gpurun_out/r5_hazard; mkdir -p $O
D=$PWD/d3human-code_amd/d3h; R=tools/probe/coresidency_repro
{
echo "== sharing build, small mesh"; $R $D/libd3h_share.so 30 50000 700 2 0
echo "== sharing build, config-3 mesh, chain capped to 131 CUs"; $R $D/libd3h_share.so 30 50000 8770 4 131
echo "== control: exact-f32 dW kernel in its place (D3H_DW_X3=0)"; D3H_DW_X3=0 $R $D/libd3h_share.so 30 50000 8770 4 131
echo "== control: no chain at all"; REPRO_NO_CHAIN=1 $R $D/libd3h_share.so 10 50000 8770 4 131
for v in share_noput share_nomfma share_noflush; do [ -f $D/libd3h_$v.so ] && { echo "== variant $v"; $R $D/libd3h_$v.so 30 50000 8770 4 131; }; done
echo "== default build (claims the register file)"; $R $D/libd3h_hip.so 30 50000 8770 4 131
} > $O/repro_v2.txt 2>&1
tail -c 9000 $O/repro_v2.txt
