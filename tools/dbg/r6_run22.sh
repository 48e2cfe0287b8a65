REPO=$GRAFT_REPO_ROOT
mkdir -p $REPO/gpurun_out/r6
cd /tmp && export TMPDIR=/tmp
rm -f $REPO/gpurun_out/r6/ssim_kstats3.txt
for cfg in "A=1" "D3H_SSIM_DEBUG_SKIP_ALL=1"; do
  rm -rf /tmp/kp; env $cfg PROBE_OCC=1 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/kp -o r -- python3 $REPO/tools/dbg/gpu_ssim_occ_probe.py > /tmp/kp.log 2>&1
  echo "== $cfg" >> $REPO/gpurun_out/r6/ssim_kstats3.txt
  f=$(find /tmp/kp -name "*kernel_stats.csv" | head -1)
  python3 - "$f" >> $REPO/gpurun_out/r6/ssim_kstats3.txt <<'PY'
import csv,sys
for r in csv.DictReader(open(sys.argv[1])):
    n=r['Name']
    if any(k in n for k in ('ssim',)):
        print('%-60s calls %s avg %.1f us'%(n.replace('(anonymous namespace)::','')[:60], r['Calls'], float(r['AverageNs'])/1e3))
PY
done
