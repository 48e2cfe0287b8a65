#!/bin/bash
# torch-free reproducer + host-side discriminators for the x3 co-residency hazard.
#   bash tools/build_variant.sh share -DD3H_DWX_SHARE_SIMDS; hipcc ... tools/probe/coresidency_repro.cpp; gpurun -- 'bash tools/dbg/hazard_experiments.sh'
cd ${GRAFT_REPO_ROOT:-.}
SH=$PWD/d3human-code_amd/d3h/libd3h_share.so
O=gpurun_out/r5_hazard; mkdir -p $O
{
echo "== 1. torch-free reproducer (tools/probe/coresidency_repro.cpp), sharing build"
timeout 300 tools/probe/coresidency_repro $SH 40 50000 700 2 0
timeout 300 tools/probe/coresidency_repro $SH 40 50000 700 2 131
timeout 300 tools/probe/coresidency_repro $SH 40 50000 8770 4 131
timeout 300 tools/probe/coresidency_repro $SH 40 6250 700 2 0
echo "== 1b. the same with the default build (the kernel claims the register file)"
timeout 300 tools/probe/coresidency_repro $PWD/d3human-code_amd/d3h/libd3h_hip.so 20 50000 700 2 0
echo "== 2. the Python two-tick script, sharing build (ticks of 96 that differ from the first)"
run() { echo "-- $1"; env $1 D3H_LIB_PATH=$SH NO_PG=1 timeout 600 python tools/dbg/gpu_dbg_x3_race.py 48 > $O/race_$2.log 2>&1; echo "   differing ticks: $(grep -c 'bad: \[(' $O/race_$2.log) of 96; rc=$?"; grep -m3 "first differing\|h:verts rows\|deform rows" $O/race_$2.log; }
run "HOOKS=1" base
run "HOOKS=1 PYTORCH_NO_CUDA_MEMCACHING=1" nocache
run "HOOKS=1 D3H_EIK_FULL_JOIN=1" fulljoin
run "HOOKS=1 D3H_EARLY_EIKONAL=0" noearly
run "HOOKS=1 D3H_EIK_SPLIT_ISSUE=0" nosplit
run "HOOKS=1 D3H_ASYNC_TABLE_GRAD=0" notablestream
run "HOOKS=1 TRACE=1" trace
} 2>&1 | tee $O/summary.txt
