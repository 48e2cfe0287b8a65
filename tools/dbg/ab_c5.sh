run() { env $1 python bench.py --steps 60 --warmup 10 --no-cpu-baseline --no-extras $2 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print('%.2f it/s  %.3f ms' % (d['value'], d['ms_per_step']))"; }
python -m pytest tests -m gpu -x -q 2>&1 | tail -2
for i in 1 2; do
  for C in "--config 5" "--config 6" ""; do
    for E in "D3H_LPIPS_FUSED_INPUT=0" "D3H_LPIPS_FUSED_INPUT=1"; do
      echo "[cfg ${C:-3}] [$E] $(run "$E" "$C")"
    done
  done
done
