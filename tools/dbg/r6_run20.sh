mkdir -p gpurun_out/r6
REPO=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
rm -f $REPO/gpurun_out/r6/pix_kstats2.txt
for w in 1024 2048 4096 16384; do
  rm -rf /tmp/kp; PROBE_OCC=1 D3H_PIXLOSS_WGS=$w rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/kp -o r -- python3 $REPO/tools/dbg/gpu_ssim_occ_probe.py > /tmp/kp.log 2>&1
  echo "== pixel-loss workgroups $w" >> $REPO/gpurun_out/r6/pix_kstats2.txt
  f=$(find /tmp/kp -name "*kernel_stats.csv" | head -1)
  python3 - "$f" >> $REPO/gpurun_out/r6/pix_kstats2.txt <<'PY'
import csv,sys
for r in csv.DictReader(open(sys.argv[1])):
    n=r['Name']
    if any(k in n for k in ('pixel_losses',)):
        print('%-60s calls %s avg %.1f us'%(n.replace('(anonymous namespace)::','')[:60], r['Calls'], float(r['AverageNs'])/1e3))
PY
done
cd $REPO
(timeout 1500 python -m pytest tests/test_gpu_parity.py tests/test_gpu_e2e.py -m gpu -x -q 2>&1 | grep "passed\|failed\|Error\|assert" | tail -8 > gpurun_out/r6/gpu_tests_tk2.txt)
run() { echo "$1" >> gpurun_out/r6/tune_eik3.txt; for i in 1 2; do env $1 timeout 900 python bench.py --steps 100 --warmup 20 --no-cpu-baseline --no-extras --no-predict 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('   ms/step %.3f  it/s %.1f'%(d['ms_per_step'], d['value']))" >> gpurun_out/r6/tune_eik3.txt; done; }
rm -f gpurun_out/r6/tune_eik3.txt
run "D3H_NOOP=1"
run "D3H_EIK_CUS=160"
run "D3H_EIK_CUS=176"
run "D3H_SSIM_OCC=0"
run "D3H_PIXLOSS_WGS=1024"
run "D3H_NOOP=2"
