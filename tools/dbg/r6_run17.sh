mkdir -p gpurun_out/r6
rm -f gpurun_out/r6/ssim_occ_probe2.txt
for r in 8 16 32; do echo "D3H_SSIM_ROWS=$r" >> gpurun_out/r6/ssim_occ_probe2.txt; D3H_SSIM_ROWS=$r timeout 600 python tools/dbg/gpu_ssim_occ_probe.py 2>&1 | grep "^occ" >> gpurun_out/r6/ssim_occ_probe2.txt; done
(timeout 1500 python -m pytest tests/test_gpu_parity.py tests/test_gpu_e2e.py tests/test_gpu_hazard.py tests/test_gpu_data_edges.py tests/test_lpips.py -m gpu -x -q 2>&1 | grep "passed\|failed\|Error\|assert" | tail -8 > gpurun_out/r6/gpu_tests_occ2.txt)
rm -f gpurun_out/r6/bench_occ_ab2.txt
for v in "1 8" "0 32" "1 16" "1 8" "0 32" "1 16"; do set -- $v; echo "D3H_SSIM_OCC=$1 ROWS=$2" >> gpurun_out/r6/bench_occ_ab2.txt; D3H_SSIM_OCC=$1 D3H_SSIM_ROWS=$2 timeout 900 python bench.py --steps 100 --warmup 20 --no-cpu-baseline --no-extras --no-predict 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('   ms/step %.3f  it/s %.1f'%(d['ms_per_step'], d['value']))" >> gpurun_out/r6/bench_occ_ab2.txt; done
