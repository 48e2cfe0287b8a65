#!/bin/bash
# tuning A/Bs of the side-stream budget after the compact backward changed the balance of the two streams
cd ${GRAFT_REPO_ROOT:-.}
O=gpurun_out/r5; mkdir -p $O
run() { echo -n "$1: "; env $1 python bench.py --steps 100 --warmup 10 --no-cpu-baseline --no-extras 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print('%.2f it/s  %.3f ms' % (d['value'], d['ms_per_step']))"; }
{
for r in 1 2; do
run "D3H_EIK_CUS=131"
run "D3H_EIK_CUS=0"
run "D3H_EIK_CUS=196"
run "D3H_EIK_CUS=160"
run "D3H_EIK_CUS=96"
run "D3H_DWX_DUAL_SPLIT=16"
run "D3H_DWX_DUAL_SPLIT=21"
run "D3H_EARLY_EIKONAL=0"
done
} > $O/ab_tuning.txt 2>&1
cat $O/ab_tuning.txt
