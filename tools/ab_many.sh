#!/bin/bash
# bash tools/ab_many.sh <repeats> <dir>... : bench.py (100 steps) in every checkout in turn, on one box
R=$1; shift
for i in $(seq 1 $R); do
  for D in "$@"; do
    L=$(cd $D && python bench.py --steps 100 --warmup 10 --no-cpu-baseline --no-extras 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print('%.2f it/s  %.3f ms' % (d['value'], d['ms_per_step']))")
    echo "[$D] $L"
  done
done
