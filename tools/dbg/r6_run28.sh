REPO=$GRAFT_REPO_ROOT
mkdir -p $REPO/gpurun_out/r6
(timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "ssim or pixel" 2>&1 | grep "passed\|failed\|Error\|assert" | tail -4 > gpurun_out/r6/gpu_tests_28.txt)
cd /tmp && export TMPDIR=/tmp
for cfg in "1 32" "1 16" "1 8" "0 32"; do set -- $cfg
  rm -rf /tmp/kp; PROBE_OCC=$1 D3H_SSIM_ROWS=$2 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/kp -o r -- python3 $REPO/tools/dbg/gpu_ssim_occ_probe.py > /tmp/kp.log 2>&1
  echo "== occ $1 rows $2" >> $REPO/gpurun_out/r6/gpu_tests_28.txt
  f=$(find /tmp/kp -name "*kernel_stats.csv" | head -1)
  python3 - "$f" >> $REPO/gpurun_out/r6/gpu_tests_28.txt <<'PY'
import csv,sys
for r in csv.DictReader(open(sys.argv[1])):
    n=r['Name']
    if any(k in n for k in ('ssim',)):
        print('%-60s calls %s avg %.1f us'%(n.replace('(anonymous namespace)::','')[:60], r['Calls'], float(r['AverageNs'])/1e3))
PY
done
cd $REPO
run() { echo "$1" >> gpurun_out/r6/gpu_tests_28.txt; env $1 timeout 900 python bench.py --steps 100 --warmup 20 --no-cpu-baseline --no-extras --no-predict 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('   ms/step %.3f  it/s %.1f'%(d['ms_per_step'], d['value']))" >> gpurun_out/r6/gpu_tests_28.txt; }
for i in 1 2 3; do run "D3H_SSIM_OCC=1"; run "D3H_SSIM_OCC=0"; run "D3H_SSIM_ROWS=16"; done
