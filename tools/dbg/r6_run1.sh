mkdir -p gpurun_out/r6
(timeout 2400 python tools/dbg/gpu_parity_rootcause.py 0,1,2,3 > gpurun_out/r6/rootcause_a.txt 2>&1; echo "rc $?" >> gpurun_out/r6/rootcause_a.txt)
(timeout 1500 python -m pytest tests -m gpu -q -x -s 2>&1 | grep -v Warning | tail -60 > gpurun_out/r6/gpu_suite_1.txt)
