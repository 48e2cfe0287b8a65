#!/bin/bash
# Variant build of libd3h_hip.so for A/B experiments: bash tools/build_variant.sh <name> [-DFLAG=..]...  -> d3human-code_amd/d3h/libd3h_<name>.so
NAME=$1; shift
ROOT=$(cd $(dirname $0)/.. && pwd)
O=/tmp/d3h_variant_$NAME; mkdir -p $O
for s in $ROOT/d3human-code_amd/csrc/*.hip; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -fPIC -std=c++17 -Wno-unused-value -Wno-pass-failed "$@" -c $s -o $O/$(basename $s .hip).o -I $ROOT/d3human-code_amd/csrc &
done
wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $ROOT/d3human-code_amd/d3h/libd3h_$NAME.so $O/*.o && echo built libd3h_$NAME.so
