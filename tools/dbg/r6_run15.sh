mkdir -p gpurun_out/r6
(timeout 1500 python -m pytest tests/test_gpu_parity.py tests/test_gpu_e2e.py tests/test_gpu_hazard.py tests/test_gpu_data_edges.py tests/test_lpips.py -m gpu -x -q 2>&1 | grep "passed\|failed\|Error\|assert" | tail -8 > gpurun_out/r6/gpu_tests_occ.txt)
rm -f gpurun_out/r6/bench_occ_ab.txt
for v in 1 0 1 0 1 0; do echo "D3H_SSIM_OCC=$v" >> gpurun_out/r6/bench_occ_ab.txt; D3H_SSIM_OCC=$v D3H_BENCH_DETAIL=gpurun_out/r6/detail_occ_$v.json timeout 900 python bench.py --steps 100 --warmup 20 --no-cpu-baseline --no-extras --no-predict 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('   ms/step %.3f  it/s %.1f'%(d['ms_per_step'], d['value']))" >> gpurun_out/r6/bench_occ_ab.txt; done
python - <<'PY' >> gpurun_out/r6/bench_occ_ab.txt
import json
for v in (1,0):
    d=json.load(open('gpurun_out/r6/detail_occ_%d.json'%v))
    for r in d['rooflines']:
        if 'ssim' in r.get('kernel','') or 'pixel_losses' in r.get('kernel',''):
            print(v, r['kernel'][:40], round(r['launch_ms']*1e3,1),'us')
PY
