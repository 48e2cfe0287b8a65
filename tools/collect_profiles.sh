#!/bin/bash
# Everything profiles/ holds for a round, produced on the GPU box in one go:  gpurun -- 'bash tools/collect_profiles.sh r2'
#   <tag>_bench_config3.json                     the bench.py stdout line (no profiler attached): value, roofline, cpu_baseline (< 4 KB)
#   <tag>_bench_config3_detail.json              the full record of the same run (rooflines, predicted_scaling, parity, per-stage CPU seconds)
#   <tag>_bench_config3_kernel_stats.csv         rocprofv3 --kernel-trace --stats summary of the same command
#   <tag>_bench_config3_per_iteration.csv        per-iteration launches / busy us (tools/trace_window.py), concurrent streams
#   <tag>_bench_config3_per_iteration_serialised.csv   the same with every kernel timed alone (D3H_NO_SIDE_STREAM=1)
#   <tag>_bench_config3_timeline.csv             every launch of the last iteration
#   <tag>_pmc_fetch_write.csv                    FETCH_SIZE / WRITE_SIZE per dispatch (separate passes)
#   <tag>_pmc_mfma_busy.csv                      SQ_VALU_MFMA_BUSY_CYCLES, SQ_BUSY_CYCLES, SQ_WAVE_CYCLES, SQ_INSTS_VALU_MFMA_MOPS_F32... of the SDF kernels
TAG=${1:-r3}
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/profiles_$TAG
mkdir -p $OUT
# (the stdout line is the short result line the driver parses; the full record -- rooflines, predicted_scaling, parity -- is the detail file)
cd $REPO && D3H_BENCH_DETAIL=$OUT/${TAG}_bench_config3_detail.json python3 bench.py --steps 100 --warmup 10 > $OUT/${TAG}_bench_config3.json 2> $OUT/bench.err
bash tools/profile_bench.sh $TAG --no-extras
for k in kernel_stats per_iteration per_iteration_serialised timeline; do cp gpurun_out/prof_${TAG}_$k.csv $OUT/${TAG}_bench_config3_$k.csv; done
cd /tmp && export TMPDIR=/tmp
i=0
for C in "FETCH_SIZE" "WRITE_SIZE"; do
  i=$((i+1)); rm -rf /tmp/pmcfw$i
  rocprofv3 --pmc $C --kernel-trace --output-format csv -d /tmp/pmcfw$i -o r -- python3 $REPO/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-extras > /tmp/pmcfw$i.log 2>&1
done
python3 $REPO/tools/pmc_summary.py $OUT/${TAG}_pmc_fetch_write.csv $(find /tmp/pmcfw1 /tmp/pmcfw2 -name '*counter_collection.csv')
i=0
for C in "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAVES" "SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY" "SQ_INSTS_MFMA SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_LDS"; do
  i=$((i+1)); rm -rf /tmp/pmcm$i
  D3H_NO_SIDE_STREAM=1 D3H_ASYNC_TABLE_GRAD=0 rocprofv3 --pmc $C --kernel-trace --output-format csv -d /tmp/pmcm$i -o r -- python3 $REPO/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-extras > /tmp/pmcm$i.log 2>&1
done
python3 - <<PY
import csv, glob, collections
acc = collections.defaultdict(list)
for f in glob.glob('/tmp/pmcm*/**/*counter_collection.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        n = r['Kernel_Name']
        if 'sdf_mlp' in n or 'texmlp' in n:
            acc[(n[:100], r['Counter_Name'])].append(float(r['Counter_Value']))
with open('$OUT/${TAG}_pmc_mfma_busy.csv', 'w') as fh:
    fh.write('# rocprofv3 --pmc (three passes, serialised streams: D3H_NO_SIDE_STREAM=1) -- python3 bench.py --steps 3 --warmup 1; mean per dispatch\n')
    fh.write('# MFMA utilisation of a kernel = SQ_VALU_MFMA_BUSY_CYCLES / SQ_BUSY_CYCLES (both summed over the SQs that ran it)\n')
    w = csv.writer(fh)
    w.writerow(['kernel', 'counter', 'dispatches', 'mean', 'max'])
    for (k, c), v in sorted(acc.items()):
        w.writerow([k, c, len(v), f'{sum(v) / len(v):.0f}', f'{max(v):.0f}'])
    # kernel durations of the same (first) counter pass, from its kernel trace
    dur = collections.defaultdict(list)
    for f in glob.glob('/tmp/pmcm1/**/*kernel_trace.csv', recursive=True):
        for r in csv.DictReader(open(f)):
            dur[r['Kernel_Name'][:100]].append(int(r['End_Timestamp']) - int(r['Start_Timestamp']))
    names = sorted({k for k, _ in acc})
    fh.write('# ---- derived: matrix-pipe utilisation = SQ_VALU_MFMA_BUSY_CYCLES / (1024 SIMDs x kernel duration x 2.4 GHz) -- the counter sums the\n')
    fh.write('# busy cycles of all 1024 matrix pipes (= MFMA count x 32 cycles for v_mfma_f32_16x16x4_f32); durations from the kernel trace of the same pass\n')
    for k in names:
        m = acc.get((k, 'SQ_VALU_MFMA_BUSY_CYCLES'))
        if m and dur.get(k) and sum(m) > 0:
            d_ns = sum(dur[k]) / len(dur[k])
            fh.write(f'# {k[:90]}: mean {d_ns / 1e3:.1f} us, MFMA busy {sum(m) / len(m):.3e} cycles -> utilisation {sum(m) / len(m) / (1024 * d_ns * 2.4):.3f}\n')
PY
ls -la $OUT
