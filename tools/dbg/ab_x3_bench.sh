#!/bin/bash
# config-3 step time with variant builds of the library (tools/build_variant.sh), alternated on one box
for rep in 1 2; do
for v in "$@"; do
  D3H_LIB_PATH=$PWD/d3human-code_amd/d3h/libd3h_$v.so python bench.py --no-cpu-baseline --no-extras --no-predict --steps 150 --warmup 20 2>/dev/null | tail -1 | \
     python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$v rep $rep', round(d['value'],2), round(d['ms_per_step'],3), 'sweep ms', round(d['roofline']['launch_ms'],4))"
done
done
