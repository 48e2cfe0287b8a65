export D3H_NO_SIDE_STREAM=1
for i in 1 2; do
for V in aa1 aa2 aa4 aa8; do
 D3H_LIB_PATH=d3human-code_amd/d3h/libd3h_$V.so python bench.py --steps 60 --warmup 8 --no-cpu-baseline --no-extras 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.readline())
r = {x['kernel'].split('_kernel')[0]: x['launch_ms'] for x in d['rooflines']}
print('[$V] %.2f it/s %.3f ms | aa_fwd=%.0f aa_bwd=%.0f composite_fwd=%.0f' % (d['value'], d['ms_per_step'], 1e3*r.get('aa_fwd',0), 1e3*r.get('aa_bwd',0), 1e3*r.get('composite_fwd',0)))"
done
done
python -m pytest tests/test_gpu_parity.py tests/test_gpu_e2e.py -x -q 2>&1 | grep "passed\|failed"
