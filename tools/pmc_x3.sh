# SQ / LDS / TCC counters of the SDF forward sweeps (exact-f32 and x3), stand-alone:   gpurun -- 'bash tools/pmc_x3.sh > gpurun_out/pmc_x3.txt'
cd /tmp && export TMPDIR=/tmp
i=0
for C in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_BF16" "SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INST_CYCLES_VMEM" "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_EA0_RDREQ_sum" "SQ_INSTS_VMEM SQ_ACTIVE_INST_VMEM SQ_WAIT_INST_ANY SQ_INSTS_SALU SQ_ACTIVE_INST_SCA SQ_INSTS_VALU_MFMA_F32 SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_FLAT"; do
  i=$((i+1)); rm -rf /tmp/pmcx$i
  rocprofv3 --pmc $C --kernel-trace --output-format csv -d /tmp/pmcx$i -o r -- python3 $GRAFT_REPO_ROOT/tools/gpu_probe_x3.py > /tmp/pmcx$i.log 2>&1; grep -E "Error|error|Traceback" /tmp/pmcx$i.log | head -3
done
cd $GRAFT_REPO_ROOT
python3 - <<PY
import csv, glob, collections
acc = collections.defaultdict(list)
for f in glob.glob('/tmp/pmcx*/**/*counter_collection.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        n = r['Kernel_Name']
        if 'sdf_mlp_fwd' in n:
            key = ('x3' if 'x3' in n else 'f32') + (' <false,0>' if 'false, 0' in n or 'ELb0ELi0' in n else ' small')
            acc[(key, r['Counter_Name'])].append(float(r['Counter_Value']))
last = None
for (k, c), v in sorted(acc.items()):
    if k != last: print('==', k, 'dispatches', len(v)); last = k
    v = sorted(v)
    print(f'   {c:45s} mean {sum(v)/len(v):16.0f}   max {v[-1]:16.0f}')
PY
