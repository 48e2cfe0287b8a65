#!/bin/bash
# kernel timeline of the reproducer: which kernel of the chain overlaps the victim launches that come out wrong?
cd ${GRAFT_REPO_ROOT:-.}
O=$PWD/gpurun_out/r5_hazard; mkdir -p $O
R=$PWD/tools/probe/coresidency_repro; S=$PWD/d3human-code_amd/d3h/libd3h_share.so
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/hz_trace
REPRO_ONLY="no atomics" rocprofv3 --kernel-trace --output-format csv -d /tmp/hz_trace -o r -- $R $S 4 50000 8770 4 131 > $O/trace_run.txt 2>&1
T=$(find /tmp/hz_trace -name '*kernel_trace.csv' | head -1)
python3 - "$T" $O/trace_run.txt > $O/trace_summary.txt <<'PY'
import csv, sys, re
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
t0 = int(rows[0]['Start_Timestamp'])
short = lambda n: re.sub(r'\(.*', '', n.replace('(anonymous namespace)::', '').replace('d3h_sdf::', ''))[:60]
# rounds: each starts with the chain's first kernel (the tangent sweep = sdf_mlp_fwd_x3_kernel<true...) after the setup phase
chain = [r for r in rows if 'lbs_bwd' not in r['Kernel_Name'] and 'sum_frames' not in r['Kernel_Name']]
vict = [r for r in rows if 'lbs_bwd_kernel' in r['Kernel_Name']]
print('kernels:', len(rows), 'victim launches:', len(vict))
# print the last round in full: chain kernels with times, and the victims overlapping each
dw = [r for r in chain if 'dw_layers_x3' in r['Kernel_Name']]
for k, d in enumerate(dw):
    s, e = int(d['Start_Timestamp']), int(d['End_Timestamp'])
    # the chain kernels of this round = those within 5 ms before d
    print(f'--- round with dW x3 kernel #{k}: dW {((s - t0) / 1e3):.1f} .. {((e - t0) / 1e3):.1f} us')
    for c in chain:
        cs, ce = int(c['Start_Timestamp']), int(c['End_Timestamp'])
        if s - 4_000_000 < cs < e + 1_000_000:
            print(f'   chain  {(cs - t0) / 1e3:10.1f} .. {(ce - t0) / 1e3:10.1f} us  {short(c["Kernel_Name"])}  grid {c.get("Grid_Size_X", "?")}x{c.get("Grid_Size_Y", "?")}x{c.get("Grid_Size_Z", "?")} wg {c.get("Workgroup_Size_X", "?")} vgpr {c.get("VGPR_Count", c.get("Arch_VGPR_Count", "?"))} lds {c.get("LDS_Block_Size", "?")}')
    ov = [v for v in vict if int(v['Start_Timestamp']) < e and int(v['End_Timestamp']) > s]
    print(f'   victim launches overlapping the dW kernel: {len(ov)}; first/last victim index in trace order: '
          f'{vict.index(ov[0]) if ov else None} / {vict.index(ov[-1]) if ov else None}')
    for v in ov[:6]:
        print(f'      victim #{vict.index(v)}  {(int(v["Start_Timestamp"]) - t0) / 1e3:10.1f} .. {(int(v["End_Timestamp"]) - t0) / 1e3:10.1f} us')
print(open(sys.argv[2]).read()[-2500:])
PY
cat $O/trace_summary.txt | cut -c1-260 | tail -80
