mkdir -p gpurun_out/r6
(timeout 1200 python -m pytest tests/test_gpu_parity.py tests/test_gpu_e2e.py tests/test_gpu_hazard.py -m gpu -x -q 2>&1 | grep -v "Warning\|warnings.warn\|^  \|^$" | tail -6 > gpurun_out/r6/gpu_tests_jvp.txt)
for v in 1 0 1 0; do echo "D3H_SDF_H2_JVP=$v" >> gpurun_out/r6/bench_jvp_ab.txt; D3H_SDF_H2_JVP=$v D3H_BENCH_DETAIL=gpurun_out/r6/detail_jvp_$v.json timeout 900 python bench.py --steps 100 --warmup 20 --no-cpu-baseline --no-extras --no-predict 2>/dev/null | tail -1 | cut -c1-330 >> gpurun_out/r6/bench_jvp_ab.txt; done
