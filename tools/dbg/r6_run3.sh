mkdir -p gpurun_out/r6
(timeout 900 python -m pytest tests/test_gpu_e2e.py -m gpu -q -s -k "tick_split_default_path" 2>&1 | grep -v Warning | tail -60 > gpurun_out/r6/fail_split.txt)
(D3H_SDF_H2=0 timeout 900 python -m pytest tests/test_gpu_e2e.py -m gpu -q -s -k "tick_split_default_path" 2>&1 | grep -v Warning | tail -30 > gpurun_out/r6/fail_split_h2off.txt)
(timeout 1500 python -m pytest tests/test_gpu_fullsize.py -m gpu -q -s -k "whole_tick" 2>&1 | grep -v Warning | tail -80 > gpurun_out/r6/whole_tick_3.txt)
(timeout 1500 python tools/dbg/gpu_parity_rootcause.py 0 > gpurun_out/r6/rootcause_c.txt 2>&1; echo "rc $?" >> gpurun_out/r6/rootcause_c.txt)
