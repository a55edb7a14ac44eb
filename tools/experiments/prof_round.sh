#!/bin/bash
# rocprofv3 evidence for a round (argument: tag, default r02): kernel-trace stats for the seven single-GPU workloads, then
# FETCH_SIZE and WRITE_SIZE in passes of their own (never together with a trace), then a plain bench line.
TAG=${1:-r02}
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/prof_$TAG
rm -rf $O; mkdir -p $O
cd $R
W2=utt1024x10s_f32_hann400_hop160_nfft512
W1=utt1x10s_f32_hann400_hop160_nfft512
W3=flt_f64_8ch_x_1h16k_povey400_hop160_nfft512
W4=longwin_f32_8ch_x_1h48k_bh4096_hop1024_nfft4096
W6=tel1024x10s8k_f32_hann200_hop80_nfft256
W7=music64x60s44k_f32_hann1024_hop256_nfft1024
W8=music64x60s44k_f32_hann2048_hop512_nfft2048
for w in $W2 $W1 $W3 $W4 $W6 $W7 $W8; do
  S=20; [ $w = $W2 ] && S=50; [ $w = $W1 ] && S=200
  timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_$w -- python3 bench.py --workload $w --steps $S --warmup 10 --no-cpu-baseline > $O/bench_under_rocprof_$w.json 2> $O/stats_$w.err || exit 1
  timeout -s KILL 120 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/pmc_fetch_$w -- python3 bench.py --workload $w --ramp-ms 0 --steps 3 --warmup 1 --no-cpu-baseline > /dev/null 2> $O/pmc_fetch_$w.err || exit 1
  timeout -s KILL 120 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/pmc_write_$w -- python3 bench.py --workload $w --ramp-ms 0 --steps 3 --warmup 1 --no-cpu-baseline > /dev/null 2> $O/pmc_write_$w.err || exit 1
done
timeout -k 10 200 python3 bench.py --steps 50 --warmup 10 > $O/bench_plain.json 2> $O/bench_plain.err
python3 - <<PY
import csv,glob,collections,json,shutil,os
O='$O'
out={}
def stats(d):
    st=glob.glob(O+'/'+d+'/**/*kernel_stats.csv',recursive=True)
    shutil.copy(st[0], O+'/'+d+'.csv')
    return [r for r in csv.DictReader(open(st[0])) if 'stft' in r['Name']]
def pmc(d):
    f=glob.glob(O+'/'+d+'/**/*counter_collection.csv',recursive=True)[0]
    agg=collections.defaultdict(list)
    rows=[r for r in csv.DictReader(open(f)) if 'stft' in r['Kernel_Name']]
    with open(O+'/'+d+'.csv','w') as g:
        w=csv.DictWriter(g, fieldnames=['Kernel_Name','Counter_Name','Counter_Value','Grid_Size','Workgroup_Size','LDS_Block_Size','VGPR_Count','SGPR_Count'])
        w.writeheader()
        for r in rows: w.writerow({k:r[k] for k in w.fieldnames})
    for r in rows: agg[r['Counter_Name']].append(float(r['Counter_Value']))
    return {k+'_per_launch':sum(v)/len(v) for k,v in agg.items()}
for w in ('$W2','$W1','$W3','$W4','$W6','$W7','$W8'):
    out[w]={'kernel_stats':stats('stats_'+w),**pmc('pmc_fetch_'+w),**pmc('pmc_write_'+w)}
json.dump(out,open(O+'/summary.json','w'),indent=1)
for k,v in out.items(): print(k, v['kernel_stats'][0]['AverageNs'], v.get('FETCH_SIZE_per_launch'), v.get('WRITE_SIZE_per_launch'))
PY
rm -rf $O/stats_*/ $O/pmc_*/
cat $O/bench_plain.json
