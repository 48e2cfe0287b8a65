#!/bin/bash
# bash tools/ab_eik_grid.sh : CUs given to the eikonal chain on the side stream (D3H_EIK_CUS; unset = the rule of geometry/hmsdf.py:_eikonal_async)
run() { env $1 python bench.py --steps 100 --warmup 10 --no-cpu-baseline --no-extras $2 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print('%.2f it/s  %.3f ms' % (d['value'], d['ms_per_step']))"; }
for i in 1 2 3; do
  for C in "" "--config 2" "--config 5"; do
    for E in "D3H_EIK_CUS=0" "D3H_X=1"; do
      echo "[cfg ${C:-3}] [$E] $(run $E "$C")"
    done
  done
done
