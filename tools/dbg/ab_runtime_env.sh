# runtime knobs of the HIP / HSA stack against the default, config 3 (set before the process starts)
run() { env $1 python bench.py --steps 100 --warmup 10 --no-cpu-baseline --no-extras 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print('%.2f it/s  %.3f ms' % (d['value'], d['ms_per_step']))"; }
for i in 1 2 3; do
  for E in "D3H_X=1" "HIP_FORCE_DEV_KERNARG=1" "HSA_ENABLE_INTERRUPT=0" "HIP_FORCE_DEV_KERNARG=1 HSA_ENABLE_INTERRUPT=0" "GPU_MAX_HW_QUEUES=8"; do
    echo "[$E] $(run "$E")"
  done
done
