mkdir -p gpurun_out/r6
run() { echo "$1" >> gpurun_out/r6/tune_tex.txt; for i in 1 2; do env $1 timeout 900 python bench.py --steps 100 --warmup 20 --no-cpu-baseline --no-extras --no-predict 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('   ms/step %.3f  it/s %.1f'%(d['ms_per_step'], d['value']))" >> gpurun_out/r6/tune_tex.txt; done; }
rm -f gpurun_out/r6/tune_tex.txt
run "D3H_NOOP=1"
run "D3H_TEX_H2_WGS=120"
run "D3H_TEX_H2_WGS=96"
run "D3H_TEX_H2_WGS=180"
run "D3H_NOOP=2"
