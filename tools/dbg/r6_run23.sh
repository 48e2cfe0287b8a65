REPO=$GRAFT_REPO_ROOT
mkdir -p $REPO/gpurun_out/r6
(timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "ssim or pixel" 2>&1 | grep "passed\|failed\|Error\|assert" | tail -4 > gpurun_out/r6/gpu_tests_23.txt)
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/kp; PROBE_OCC=1 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/kp -o r -- python3 $REPO/tools/dbg/gpu_ssim_occ_probe.py > /tmp/kp.log 2>&1
f=$(find /tmp/kp -name "*kernel_stats.csv" | head -1)
python3 - "$f" >> $REPO/gpurun_out/r6/gpu_tests_23.txt <<'PY'
import csv,sys
for r in csv.DictReader(open(sys.argv[1])):
    n=r['Name']
    if any(k in n for k in ('ssim','pixel_losses')):
        print('%-60s calls %s avg %.1f us'%(n.replace('(anonymous namespace)::','')[:60], r['Calls'], float(r['AverageNs'])/1e3))
PY
