#!/bin/bash
# time the x3 forward variants built by tools/build_variant.sh (libd3h_<name>.so) on one box
for v in "$@"; do
  echo "== $v"; D3H_LIB_PATH=$PWD/d3human-code_amd/d3h/libd3h_$v.so python tools/gpu_probe_x3.py 2>&1 | grep -E "x3 .*save|n 1024"
done
