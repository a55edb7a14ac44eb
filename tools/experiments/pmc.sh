#!/bin/bash
# usage: scratch/pmc.sh <variant> <tag> "<counters>" ["<counters2>" ...]
v=$1; tag=$2; shift 2
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cd $R
if [ "$v" != "lib" ]; then export DSP_AMD_LIB=$R/scratch/variants/$v.so; fi
i=0
for c in "$@"; do
  O=$R/gpurun_out/pmc_${tag}_$i
  mkdir -p $O
  timeout -s KILL 90 rocprofv3 --pmc $c --output-format csv -d $O -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline $BARGS > /dev/null 2> $O/err.txt
  python3 - <<PY
import csv,glob,collections
fs=glob.glob('$O/**/*counter_collection.csv',recursive=True)
agg=collections.defaultdict(list)
for f in fs:
    for r in csv.DictReader(open(f)):
        if '${K:-r512}' in r['Kernel_Name']:
            agg[r['Counter_Name']].append(float(r['Counter_Value']))
            meta=(r['VGPR_Count'],r['SGPR_Count'],r['LDS_Block_Size'],r['Grid_Size'])
for k,v in sorted(agg.items()): print('$v',k,'%.4g'%(sum(v)/len(v)))
print('meta vgpr,sgpr,lds,grid',meta)
PY
  i=$((i+1))
done
