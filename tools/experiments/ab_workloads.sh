#!/bin/bash
# usage: tools/experiments/ab_workloads.sh "<workload> ..." <variant|-> ...   : bench.py's kernel time of each workload for each library
# (a name = scratch/variants/<name>.so, "-" = the library in the tree), twice round-robin
WL=$1; shift
for rep in 1 2; do
for w in $WL; do
for v in "$@"; do
  if [ "$v" = "-" ]; then unset DSP_AMD_LIB; else export DSP_AMD_LIB=$PWD/scratch/variants/$v.so; fi
  timeout -k 10 120 python bench.py --workload $w --steps 100 --warmup 100 --no-cpu-baseline 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read()); r=d['roofline']
print('$v', '$w'[:28], 'kernel_ms=%.4f' % r['kernel_ms'], 'frac=%.3f' % r['frac'])
" || { echo "$v $w FAILED"; exit 1; }
done
done
done
