# SQ / TCC / TCP counters (mean per dispatch) of the image-space and texture kernels of the serialised training step:
#   gpurun -- 'bash tools/pmc_texmlp.sh > gpurun_out/pmc_image_space.txt'
cd /tmp && export TMPDIR=/tmp
export D3H_NO_SIDE_STREAM=1
i=0
for C in "SQ_WAVES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_VMEM SQ_ACTIVE_INST_VMEM" "TCC_ATOMIC_sum TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum" "TCP_TOTAL_ATOMIC_WITHOUT_RET_sum TCP_TOTAL_READ_sum TCP_PENDING_STALL_CYCLES_sum TCP_ATOMIC_TAGCONFLICT_STALL_CYCLES_sum" "TCC_TAG_STALL_sum TCC_BUSY_sum TCP_TCC_READ_REQ_sum TCC_EA0_ATOMIC_sum"; do
  i=$((i+1)); rm -rf /tmp/pmc$i
  rocprofv3 --pmc $C --kernel-trace --output-format csv -d /tmp/pmc$i -o r -- python3 $GRAFT_REPO_ROOT/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-extras > /tmp/pmc$i.log 2>&1
done
cd $GRAFT_REPO_ROOT
python3 - <<PY
import csv, glob, collections
acc = collections.defaultdict(list)
for f in glob.glob('/tmp/pmc*/**/*counter_collection.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        n = r['Kernel_Name']
        for k in ('texmlp_fwd', 'texmlp_bwd_kernel<1>', 'texmlp_bwd_mlp', 'ssim_fwd', 'ssim_bwd', 'pixel_losses_fwd', 'pixel_losses_bwd', 'gbuffer_fwd', 'gbuffer_bwd', 'aa_fwd', 'aa_bwd',
                  'raster_bwd', 'composite_fwd', 'lbs_fwd', 'lbs_bwd', 'sdf_reg_fwd'):
            if k in n:
                acc[(k, r['Counter_Name'])].append(float(r['Counter_Value']))
last = None
for (k, c), v in sorted(acc.items()):
    if k != last: print('==', k, 'dispatches', len(v)); last = k
    print(f'   {c:45s} {sum(v)/len(v):16.0f}')
PY
