#!/bin/bash
# usage: scratch/run_ab.sh v1 v2 ...   (runs bench for each scratch/variants/<v>.so; prints kernel_ms, GB/s)
for v in "$@"; do
  DSP_AMD_LIB=$PWD/scratch/variants/$v.so timeout -k 10 120 python bench.py --steps 300 --warmup 300 --no-cpu-baseline 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read()); r=d['roofline']
print('$v', 'kernel_ms=%.4f' % r['kernel_ms'], 'GB/s=%.0f' % r['achieved'], 'frac=%.3f' % r['frac'], 'Gframes/s=%.3f' % (d['value']/1e9))
" || echo "$v FAILED"
done
