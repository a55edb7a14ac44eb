#!/bin/bash
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/wave_pmc; rm -rf $O; mkdir -p $O; cd $R
timeout -s KILL 90 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT --output-format csv -d $O/p1 -- python3 scratch/wave_case.py $1 $2 > $O/p1.log 2>&1 || exit 1
timeout -s KILL 90 rocprofv3 --pmc SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_INST_CYCLES_VMEM SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_INSTS_VMEM SQ_WAVES --output-format csv -d $O/p2 -- python3 scratch/wave_case.py $1 $2 > $O/p2.log 2>&1 || exit 1
python3 - <<PY
import csv,glob,collections
for d in ('p1','p2'):
    f=glob.glob('$O/'+d+'/**/*counter_collection.csv',recursive=True)[0]
    agg=collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        if 'wave_kernel' in r['Kernel_Name']: agg[r['Counter_Name']].append(float(r['Counter_Value']))
    for k,v in agg.items(): print(k, sum(v)/len(v))
PY
