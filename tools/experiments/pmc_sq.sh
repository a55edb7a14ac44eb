#!/bin/bash
# SQ counter passes (each in a run of its own, never with a trace) for the three benchmark kernels, or for the pairs
# <workload> <kernel-name substring> given as arguments; summaries under gpurun_out/pmc_sq.txt
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; cd $R
OUT=$R/gpurun_out/pmc_sq.txt; : > $OUT
run() {   # workload, kernel-name substring
  for c in "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR" \
           "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_INST_CYCLES_VMEM_RD SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE"; do
    O=$R/gpurun_out/pmc_sq_tmp; rm -rf $O; mkdir -p $O
    timeout -s KILL 120 rocprofv3 --pmc $c --output-format csv -d $O -- python3 bench.py --workload $1 --ramp-ms 0 --steps 3 --warmup 1 --no-cpu-baseline > /dev/null 2> $O/err.txt || { echo "FAILED $1 $c" >> $OUT; return 1; }
    python3 - "$1" "$2" >> $OUT <<PY
import csv,glob,collections,sys
agg=collections.defaultdict(list)
for f in glob.glob('$O/**/*counter_collection.csv',recursive=True):
    for r in csv.DictReader(open(f)):
        if sys.argv[2] in r['Kernel_Name']: agg[r['Counter_Name']].append(float(r['Counter_Value']))
for k,v in sorted(agg.items()): print(sys.argv[1], k, '%.6g' % (sum(v)/len(v)))
PY
  done
}
if [ $# -ge 2 ]; then
  while [ $# -ge 2 ]; do run $1 $2 || exit 1; shift 2; done
  cat $OUT; exit 0
fi
run utt1024x10s_f32_hann400_hop160_nfft512 stft_r512_fast && run flt_f64_8ch_x_1h16k_povey400_hop160_nfft512 stft_r512_fast && run longwin_f32_8ch_x_1h48k_bh4096_hop1024_nfft4096 stft_r4096
cat $OUT
