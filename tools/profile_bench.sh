#!/bin/bash
# rocprofv3 kernel trace + stats of bench.py (config 3 by default) and the per-iteration summaries; run on the GPU box through gpurun:
#   gpurun -- 'bash tools/profile_bench.sh <tag> [bench args]'   -> gpurun_out/prof_<tag>_{kernel_stats,per_iteration,per_iteration_serialised,timeline}.csv
TAG=${1:-x}; shift
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/prof_$TAG /tmp/prof_${TAG}_ser
export D3H_BENCH_DETAIL=/tmp/prof_${TAG}_detail.json
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_$TAG -o r -- python3 $REPO/bench.py --steps 20 --warmup 5 --no-cpu-baseline "$@" > /tmp/prof_$TAG.log 2>&1
D3H_NO_SIDE_STREAM=1 D3H_ASYNC_TABLE_GRAD=0 rocprofv3 --kernel-trace --output-format csv -d /tmp/prof_${TAG}_ser -o r -- python3 $REPO/bench.py --steps 20 --warmup 5 --no-cpu-baseline "$@" > /tmp/prof_${TAG}_ser.log 2>&1
cd $REPO && mkdir -p gpurun_out
T=$(find /tmp/prof_$TAG -name '*kernel_trace.csv' | head -1)
S=$(find /tmp/prof_$TAG -name '*kernel_stats.csv' | head -1)
TS=$(find /tmp/prof_${TAG}_ser -name '*kernel_trace.csv' | head -1)
cp "$S" gpurun_out/prof_${TAG}_kernel_stats.csv
python3 tools/trace_window.py "$T" gpurun_out/prof_${TAG}_per_iteration.csv --timeline gpurun_out/prof_${TAG}_timeline.csv ${ANCHOR:+--anchor $ANCHOR}
python3 tools/trace_window.py "$TS" gpurun_out/prof_${TAG}_per_iteration_serialised.csv ${ANCHOR:+--anchor $ANCHOR}
grep '^{"metric"' /tmp/prof_$TAG.log | tail -1 > gpurun_out/prof_${TAG}_bench_line.txt
