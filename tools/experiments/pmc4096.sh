#!/bin/bash
# Counter passes for the 4096-point kernel (each pass a run of its own, no tracing beside it): SQ issue / wait / LDS, then TA / TCP.
# usage: bash tools/experiments/pmc4096.sh <tag> [library]      -> gpurun_out/pmc4096_<tag>.txt
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; cd $R
TAG=$1; [ -n "$2" ] && export DSP_AMD_LIB=$2
OUT=$R/gpurun_out/pmc4096_$TAG.txt; : > $OUT
W=longwin_f32_8ch_x_1h48k_bh4096_hop1024_nfft4096
for c in "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR" \
         "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_VMEM SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" \
         "SQ_WAIT_INST_LDS SQ_INST_CYCLES_VMEM_WR SQ_INST_CYCLES_VMEM_RD SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL" \
         "TA_TA_BUSY_sum TA_BUSY_avr TCP_PENDING_STALL_CYCLES_sum TCP_TCR_TCP_STALL_CYCLES_sum TCP_TA_TCP_STATE_READ_sum TCP_GATE_EN1_sum" \
         "TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum TCP_TD_TCP_STALL_CYCLES_sum TCP_TOTAL_WRITE_sum TCP_TCC_WRITE_REQ_sum"; do
  O=$R/gpurun_out/pmc_tmp; rm -rf $O; mkdir -p $O
  timeout -s KILL 120 rocprofv3 --pmc $c --output-format csv -d $O -- python3 bench.py --workload $W --ramp-ms 0 --steps 3 --warmup 1 --no-cpu-baseline > /dev/null 2> $O/err.txt || { echo "FAILED $c: $(grep -i -m2 error $O/err.txt)" >> $OUT; continue; }
  python3 - >> $OUT <<PY
import csv,glob,collections
agg=collections.defaultdict(list)
for f in glob.glob('$O/**/*counter_collection.csv',recursive=True):
    for r in csv.DictReader(open(f)):
        if 'stft_r4096' in r['Kernel_Name']: agg[r['Counter_Name']].append(float(r['Counter_Value']))
for k,v in sorted(agg.items()): print(k, '%.6g' % (sum(v)/len(v)))
PY
done
rm -rf $R/gpurun_out/pmc_tmp
cat $OUT
