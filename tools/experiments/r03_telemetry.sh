#!/bin/bash
# Direct evidence for the shader clock inside the 4096-point kernel: the driver's own clock / power readings while the
# kernel runs back to back, and the ratio of two counters read by every wave (stamps builds), for 8 and 4 waves per CU.
set -o pipefail
O=gpurun_out/r03_telemetry; mkdir -p $O
W4=longwin_f32_8ch_x_1h48k_bh4096_hop1024_nfft4096
(which rocm-smi amd-smi; ls /sys/class/drm/ ) > $O/tools.txt 2>&1
timeout -k 10 120 python tools/experiments/clock_telemetry.py $W4 6 > $O/telemetry_4096_w8.json 2> $O/telemetry_4096_w8.err &&
DSP_AMD_LIB=$PWD/scratch/variants/w4.so timeout -k 10 120 python tools/experiments/clock_telemetry.py $W4 6 > $O/telemetry_4096_w4.json 2> $O/telemetry_4096_w4.err &&
timeout -k 10 120 python tools/experiments/clock_telemetry.py utt1024x10s_f32_hann400_hop160_nfft512 6 > $O/telemetry_512.json 2> $O/telemetry_512.err &&
DSP_AMD_LIB=$PWD/scratch/variants/stamps8.so timeout -k 10 120 python tools/experiments/stamps4.py > $O/stamps_w8.txt 2>&1 &&
DSP_AMD_LIB=$PWD/scratch/variants/stamps4.so timeout -k 10 120 python tools/experiments/stamps4.py > $O/stamps_w4.txt 2>&1
echo rc=$?
tail -5 $O/stamps_w8.txt $O/stamps_w4.txt
