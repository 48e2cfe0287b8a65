mkdir -p gpurun_out/r6
REPO=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
rm -f $REPO/gpurun_out/r6/ssim_kstats.txt
for cfg in "1 32" "0 32" "1 8"; do set -- $cfg
  rm -rf /tmp/kp; PROBE_OCC=$1 D3H_SSIM_ROWS=$2 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/kp -o r -- python3 $REPO/tools/dbg/gpu_ssim_occ_probe.py > /tmp/kp.log 2>&1
  echo "== occ $1 rows $2" >> $REPO/gpurun_out/r6/ssim_kstats.txt
  f=$(find /tmp/kp -name "*kernel_stats.csv" | head -1)
  python3 - "$f" >> $REPO/gpurun_out/r6/ssim_kstats.txt <<'PY'
import csv,sys
for r in csv.DictReader(open(sys.argv[1])):
    n=r['Name']
    if any(k in n for k in ('ssim','pixel_losses')):
        print('%-60s calls %s avg %.1f us'%(n.replace('(anonymous namespace)::','')[:60], r['Calls'], float(r['AverageNs'])/1e3))
PY
done
