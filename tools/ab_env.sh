#!/bin/bash
# bash tools/ab_env.sh <repeats> "<env A>" "<env B>" <dir>... : bench.py in every checkout under both environments, on one box
R=$1; EA="$2"; EB="$3"; shift 3
for i in $(seq 1 $R); do
  for D in "$@"; do
    for E in "$EA" "$EB"; do
      L=$(cd $D && env $E python bench.py --steps 100 --warmup 10 --no-cpu-baseline --no-extras 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print('%.2f it/s  %.3f ms' % (d['value'], d['ms_per_step']))")
      echo "[$D] [$E] $L"
    done
  done
done
