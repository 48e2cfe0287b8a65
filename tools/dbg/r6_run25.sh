mkdir -p gpurun_out/r6
run() { echo "$1" >> gpurun_out/r6/tune_eik4.txt; env $1 timeout 900 python bench.py --steps 100 --warmup 20 --no-cpu-baseline --no-extras --no-predict 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('   ms/step %.3f  it/s %.1f'%(d['ms_per_step'], d['value']))" >> gpurun_out/r6/tune_eik4.txt; }
rm -f gpurun_out/r6/tune_eik4.txt
for i in 1 2 3; do run "D3H_NOOP=1"; run "D3H_EIK_CUS=98"; run "D3H_EIK_CUS=112"; run "D3H_EIK_CUS=160"; done
