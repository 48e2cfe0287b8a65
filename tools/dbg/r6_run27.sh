mkdir -p gpurun_out/r6
(timeout 1500 python -m pytest tests/test_gpu_parity.py tests/test_gpu_e2e.py tests/test_gpu_hazard.py tests/test_gpu_data_edges.py -m gpu -x -q 2>&1 | grep "passed\|failed\|Error\|assert" | tail -8 > gpurun_out/r6/gpu_tests_27.txt)
run() { echo "$1" >> gpurun_out/r6/ab_defer.txt; env $1 timeout 900 python bench.py --steps 100 --warmup 20 --no-cpu-baseline --no-extras --no-predict 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('   ms/step %.3f  it/s %.1f'%(d['ms_per_step'], d['value']))" >> gpurun_out/r6/ab_defer.txt; }
rm -f gpurun_out/r6/ab_defer.txt
for i in 1 2 3; do run "D3H_TEX_DEFER_SCATTER=1"; run "D3H_TEX_DEFER_SCATTER=0"; done
