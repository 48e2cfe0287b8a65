#!/bin/bash
# A/B/.. of several builds on one box with the in-situ kernel times: bash tools/ab_kernels.sh <repeats> <libA.so> <libB.so> ...
R=$1; shift
for i in $(seq 1 $R); do
  for V in "$@"; do
    D3H_LIB_PATH=$V python bench.py --steps 60 --warmup 8 --no-cpu-baseline --no-extras 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.readline())
print('[%s] %.2f it/s %.3f ms | ' % ('$V'.split('/')[-1], d['value'], d['ms_per_step']) + ' '.join('%s=%.0f' % (r['kernel'].replace('sdf_mlp_', '').replace('_kernel', '')[:22].replace(' ', ''), 1e3 * r['launch_ms']) for r in d['rooflines'][:9]))"
  done
done
