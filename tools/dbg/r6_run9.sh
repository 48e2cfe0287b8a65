mkdir -p gpurun_out/r6
(timeout 1200 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fullsize.py -m gpu -x -q -k "ssim or pixel or image or config3_full or 1080" 2>&1 | grep -v "Warning\|warnings.warn\|^  \|^$" | tail -5 > gpurun_out/r6/gpu_tests_ssim.txt)
run() { echo "$1" >> gpurun_out/r6/tune_eik.txt; for i in 1 2; do env $1 timeout 900 python bench.py --steps 100 --warmup 20 --no-cpu-baseline --no-extras --no-predict 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('   ms/step %.3f  it/s %.1f'%(d['ms_per_step'], d['value']))" >> gpurun_out/r6/tune_eik.txt; done; }
run "D3H_NOOP=1"
run "D3H_EIK_CUS=98"
run "D3H_EIK_CUS=160"
run "D3H_EIK_CUS=196"
run "D3H_EIK_CUS=0"
run "D3H_DWX_DUAL_SPLIT=16"
run "D3H_DWX_DUAL_SPLIT=32"
run "D3H_NOOP=2"
