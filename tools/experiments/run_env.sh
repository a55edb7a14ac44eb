#!/bin/bash
# usage: scratch/run_env.sh <variant> "ENV=.. ENV2=.." ...
v=$1; shift
for e in "$@"; do
  env $e DSP_AMD_LIB=$PWD/scratch/variants/$v.so timeout -k 10 120 python bench.py --steps 300 --warmup 300 --no-cpu-baseline 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read()); r=d['roofline']
print('$v [$e]', 'kernel_ms=%.4f' % r['kernel_ms'], 'GB/s=%.0f' % r['achieved'], 'frac=%.3f' % r['frac'])
" || echo "$v $e FAILED"
done
