#!/bin/bash
# generic A/B of environment settings on ONE box, settings interleaved (a box drifts by 1-3 % over ten minutes: never compare blocks): tools/dbg/ab.sh <out-name> <rounds> "ENV=a" "ENV=b" ...   (bench.py config 3, 100 steps each)
cd ${GRAFT_REPO_ROOT:-.}
O=gpurun_out/ab; mkdir -p $O
NAME=$1; R=$2; shift 2
run() { echo -n "$1: "; env $1 python bench.py --steps ${STEPS:-100} --warmup 10 --no-cpu-baseline --no-extras ${BENCH_ARGS} 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print('%.2f it/s  %.3f ms' % (d['value'], d['ms_per_step']))"; }
{
for r in $(seq 1 $R); do
for e in "$@"; do run "$e"; done
done
} > $O/$NAME.txt 2>&1
cat $O/$NAME.txt
