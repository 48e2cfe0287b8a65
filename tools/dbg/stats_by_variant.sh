#!/bin/bash
# rocprofv3 --stats of a short bench run per library variant; prints the average duration of the kernels matching a pattern
#   bash tools/dbg/stats_by_variant.sh <pattern> <lib.so> ...
PAT=$1; shift
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT}
for L in "$@"; do
  export D3H_LIB_PATH=$L
  D=/tmp/p_$(basename $L .so); rm -rf $D
  rocprofv3 --kernel-trace --stats --output-format csv -d $D -o r -- python3 $R/bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extras > $D.log 2>&1
  python3 - "$PAT" "$(find $D -name '*kernel_stats.csv' | head -1)" "$(basename $L)" <<'PY'
import csv, sys, re
pat, path, lib = sys.argv[1:4]
for r in csv.DictReader(open(path)):
    if re.search(pat, r['Name']):
        print(f"{lib:22s} {r['Name'][:70]:70s} calls {r['Calls']:>6} avg_us {float(r['AverageNs'])/1e3:9.1f}")
PY
done
