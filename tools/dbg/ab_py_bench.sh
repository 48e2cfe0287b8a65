#!/bin/bash
# A/B of two CHECKOUTS' Python layers is not possible on one box; this alternates env switches instead:  tools/dbg/ab_py_bench.sh "VAR=a" "VAR=b"
for rep in 1 2 3; do
for v in "$@"; do
  env $v python bench.py --no-cpu-baseline --no-extras --no-predict --steps 150 --warmup 20 2>/dev/null | tail -1 | \
     python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$v rep $rep', round(d['value'],2), round(d['ms_per_step'],3))"
done
done
