mkdir -p gpurun_out/r6
(timeout 1800 python -m pytest tests/test_gpu_parity.py tests/test_gpu_e2e.py tests/test_gpu_hazard.py tests/test_gpu_data_edges.py tests/test_lpips.py tests/test_bench_launch.py -m gpu -x -q 2>&1 | grep "passed\|failed\|Error\|assert" | tail -8 > gpurun_out/r6/gpu_tests_21.txt)
bash tools/profile_bench.sh r6b --no-extras > /dev/null 2>&1
python - <<'PY' > gpurun_out/r6/serialised_21.txt
import csv
rows=list(csv.DictReader(open('gpurun_out/prof_r6b_per_iteration_serialised.csv')))
tot=0
for r in rows:
    try: b=float(r['BusyUsPerIter'])
    except: continue
    tot+=b
print('total', tot)
for r in rows[:40]:
    n=r['Name'].replace('(anonymous namespace)::','').replace('void ','')
    print('%-62s %6s %8s %8s'%(n[:62], r['LaunchesPerIter'], r['BusyUsPerIter'], r['AvgUs']))
PY
for i in 1 2 3; do timeout 900 python bench.py --steps 100 --warmup 20 --no-cpu-baseline --no-extras --no-predict 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('   ms/step %.3f  it/s %.1f'%(d['ms_per_step'], d['value']))" >> gpurun_out/r6/serialised_21.txt; done
