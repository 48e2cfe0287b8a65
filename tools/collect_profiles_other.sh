#!/bin/bash
# The other configurations of a round:  gpurun -- 'bash tools/collect_profiles_other.sh r3'
#   <tag>_bench_config{2,5,6,f3c}.json, <tag>_bench_config3_all12.json    the bench.py lines (no profiler attached)
#   <tag>_bench_config5_per_iteration{,_serialised}.csv, <tag>_bench_config3_all12_per_iteration{,_serialised}.csv   tools/trace_window.py
TAG=${1:-r3}
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/profiles_$TAG
mkdir -p $OUT
cd $REPO
for C in 2 5 6 f3c; do D3H_BENCH_DETAIL=$OUT/${TAG}_bench_config${C}_detail.json python3 bench.py --config $C --no-cpu-baseline --no-extras > $OUT/${TAG}_bench_config$C.json 2> $OUT/bench_c$C.err; done
D3H_BENCH_DETAIL=$OUT/${TAG}_bench_config3_all12_detail.json python3 bench.py --all-buffers --no-cpu-baseline --no-extras > $OUT/${TAG}_bench_config3_all12.json 2> $OUT/bench_all12.err
ANCHOR=adam_multi_kernel bash tools/profile_bench.sh ${TAG}c5 --config 5 --no-extras
for k in per_iteration per_iteration_serialised; do cp gpurun_out/prof_${TAG}c5_$k.csv $OUT/${TAG}_bench_config5_$k.csv; done
bash tools/profile_bench.sh ${TAG}a12 --all-buffers --no-extras
for k in per_iteration per_iteration_serialised; do cp gpurun_out/prof_${TAG}a12_$k.csv $OUT/${TAG}_bench_config3_all12_$k.csv; done
ls -la $OUT
