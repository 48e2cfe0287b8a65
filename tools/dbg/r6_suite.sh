# one full pass of the GPU suite exactly as the driver runs it (-x), tail kept
mkdir -p gpurun_out/r6
TAG=${1:-0}
(timeout 2400 python -m pytest tests -m gpu -x -q 2>&1 | grep -v "Warning\|warnings.warn\|^  \|^$" | tail -25 > gpurun_out/r6/gpu_suite_x_$TAG.txt)
