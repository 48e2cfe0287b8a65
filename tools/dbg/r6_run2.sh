mkdir -p gpurun_out/r6
(timeout 900 python tools/gpu_probe_x3.py > gpurun_out/r6/probe_h2.txt 2>&1; echo "rc $?" >> gpurun_out/r6/probe_h2.txt)
(timeout 1500 python tools/dbg/gpu_parity_rootcause.py 0,1 > gpurun_out/r6/rootcause_b.txt 2>&1; echo "rc $?" >> gpurun_out/r6/rootcause_b.txt)
for h in 1 0 1 0; do D3H_SDF_H2=$h timeout 900 python bench.py --steps 100 --warmup 20 --no-cpu-baseline --no-extras --no-predict 2>gpurun_out/r6/bench_h2_$h.err | tail -1 | cut -c1-600 >> gpurun_out/r6/bench_h2_ab.txt; done
(timeout 2400 python -m pytest tests -m gpu -q -s 2>&1 | grep -v Warning | tail -80 > gpurun_out/r6/gpu_suite_2.txt)
