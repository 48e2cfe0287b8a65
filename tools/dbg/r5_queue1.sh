#!/bin/bash
cd ${GRAFT_REPO_ROOT:-.}
O=gpurun_out/r5; mkdir -p $O
D=$PWD/d3human-code_amd/d3h
(timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "rasterize" 2>&1 | tail -5) > $O/q1_raster_tests.txt
timeout 600 python tools/gpu_probe_raster_tris.py 2>&1 | grep -v amdgpu.ids > $O/raster_vs_triangles.txt
for i in 1 2; do for V in hip pipe; do echo -n "$V: "; D3H_LIB_PATH=$D/libd3h_$V.so python bench.py --steps 100 --warmup 10 --no-cpu-baseline --no-extras 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.readline()); dd = json.load(open('bench_detail.json'))
dw = [r for r in dd['rooflines'] if 'dw_layers' in r['kernel']]
print('%.2f it/s  %.3f ms  sweep %.3f ms | ' % (d['value'], d['ms_per_step'], d['roofline']['launch_ms']) + '  '.join('%s %.0f us' % (r['kernel'][-24:], 1e3 * r['launch_ms']) for r in dw))"; done; done > $O/ab_pipe.txt 2>&1
(D3H_LIB_PATH=$D/libd3h_pipe.so timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "sdf_mlp" 2>&1 | tail -4) > $O/q1_pipe_tests.txt
timeout 1500 python tools/dbg/gpu_parity_bars.py 6 c3 > gpurun_out/r5_parity_bars.txt 2>&1
cat $O/q1_raster_tests.txt $O/raster_vs_triangles.txt $O/ab_pipe.txt $O/q1_pipe_tests.txt
