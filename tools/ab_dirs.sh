#!/bin/bash
# A/B of two checkouts on ONE box: bash tools/ab_dirs.sh <dirA> <dirB> [repeats] [bench args...]; prints it/s of every run, alternating
A="$1"; B="$2"; R=${3:-3}; shift 3
for i in $(seq 1 $R); do
  for D in "$A" "$B"; do
    L=$(cd $D && python bench.py --steps 100 --warmup 10 --no-cpu-baseline --no-extras "$@" 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print('%.2f it/s  %.3f ms  sdf_fwd %.3f ms' % (d['value'], d['ms_per_step'], d['roofline']['launch_ms']))")
    echo "[$D] $L"
  done
done
