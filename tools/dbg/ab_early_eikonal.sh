#!/bin/bash
# A/B of D3H_EARLY_EIKONAL on one box: config 3, 2, f3c; alternated to cancel drift.
mkdir -p gpurun_out/ab
for rep in 1 2; do
  for cfg in 3 2 f3c; do
    for v in 1 0; do
      D3H_EARLY_EIKONAL=$v python bench.py --config $cfg --no-cpu-baseline --no-extras --no-predict --steps 200 --warmup 30 2>/dev/null | tail -1 | \
        python -c "import sys,json; d=json.loads(sys.stdin.read()); print('cfg $cfg early=$v rep $rep', d['value'], d['ms_per_step'])" | tee -a gpurun_out/ab/early.log
    done
  done
done
