#!/bin/bash
# how often does the plain-vs-arena tick comparison of test_gpu_rccl_collectives_single_rank fail, per environment switch?
for env in "A=1" "D3H_SDF_X3=0" "D3H_EARLY_EIKONAL=0" "D3H_NO_SIDE_STREAM=1"; do
  f=0
  for i in 1 2 3 4 5 6 7 8; do
    env $env python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k rccl_collectives 2>&1 | grep -q "1 passed" || f=$((f+1))
  done
  echo "$env: $f of 8 failed"
done
