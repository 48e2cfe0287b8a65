#!/bin/bash
# A/B on one box: bash tools/ab_bench.sh "<env A>" "<env B>" [repeats] [bench args...]; prints it/s of every run
A="$1"; B="$2"; R=${3:-3}; shift 3
for i in $(seq 1 $R); do
  for V in "$A" "$B"; do
    L=$(env $V python bench.py --steps 100 --warmup 10 --no-cpu-baseline "$@" 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print('%.2f it/s  %.3f ms  sdf_fwd %.3f ms' % (d['value'], d['ms_per_step'], d['roofline']['launch_ms']))")
    echo "[$V] $L"
  done
done
