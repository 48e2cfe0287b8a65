mkdir -p gpurun_out/r6
(timeout 900 python tools/gpu_probe_x3.py 2>&1 | grep -v Warning > gpurun_out/r6/probe_h2_double.txt)
(timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fullsize.py -m gpu -x -q -k "sdf or sweep or x3 or whole_tick or config3" 2>&1 | grep -v "Warning\|warnings.warn\|^  \|^$" | tail -8 > gpurun_out/r6/gpu_tests_double.txt)
for i in 1 2; do timeout 900 python bench.py --steps 100 --warmup 20 --no-cpu-baseline --no-extras --no-predict 2>/dev/null | tail -1 | cut -c1-330 >> gpurun_out/r6/bench_double.txt; done
