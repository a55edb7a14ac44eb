#!/bin/bash
# Everything the latest files under profiles/ come from, in one call on the GPU box (outputs under gpurun_out/):
# bash tools/experiments/r03_final.sh
set -o pipefail
R=$GRAFT_REPO_ROOT; cd $R; mkdir -p gpurun_out
bash tools/experiments/prof_round.sh r03 > gpurun_out/prof_r03.log 2>&1 || { tail -n 20 gpurun_out/prof_r03.log; exit 1; }
tail -n 9 gpurun_out/prof_r03.log
bash tools/experiments/prof_f64.sh > gpurun_out/prof_f64.log 2>&1 || { tail -n 20 gpurun_out/prof_f64.log; exit 1; }
timeout -k 10 300 python tools/experiments/time_canonical.py > gpurun_out/r03_canonical_overlaps.txt 2>&1 || exit 1
cat gpurun_out/r03_canonical_overlaps.txt
timeout -k 10 200 python tools/experiments/time_long_nfft.py 10 4096 > gpurun_out/r03_long_nfft.txt 2>&1 && timeout -k 10 200 python tools/experiments/time_long_nfft.py 10 8192 >> gpurun_out/r03_long_nfft.txt 2>&1 || exit 1
cat gpurun_out/r03_long_nfft.txt
timeout -k 10 200 python tools/experiments/time_hop_gt_512.py > gpurun_out/r03_hop_gt_512.txt 2>&1 || exit 1
cat gpurun_out/r03_hop_gt_512.txt
cd /tmp && export TMPDIR=/tmp PYTHONPATH=$R
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_canon -- python3 $R/tools/experiments/time_canonical.py > /dev/null 2> $R/gpurun_out/prof_canon.err || exit 1
cd $R
cp $(find gpurun_out/prof_canon -name "*kernel_stats.csv" | head -n 1) gpurun_out/r03_kernel_stats_canonical_overlaps.csv && rm -rf gpurun_out/prof_canon
