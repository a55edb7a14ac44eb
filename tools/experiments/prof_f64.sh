#!/bin/bash
# kernel-trace stats and FETCH_SIZE / WRITE_SIZE passes (each in a run of its own) for the float64 workloads of the
# 64 / 128 / 256 and 1024 / 2048 kernels; tools/experiments/collect_f64.py merges the result into profiles/
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/prof_f64
rm -rf $O; mkdir -p $O
cd $R
for w in tel1024x10s8k_f64_hann200_hop80_nfft256 music64x60s44k_f64_hann1024_hop256_nfft1024 music64x60s44k_f64_hann2048_hop512_nfft2048; do
  timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_$w -- python3 bench.py --workload $w --steps 20 --warmup 10 --no-cpu-baseline > $O/bench_under_rocprof_$w.json 2> $O/stats_$w.err || exit 1
  timeout -s KILL 120 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/pmc_fetch_$w -- python3 bench.py --workload $w --ramp-ms 0 --steps 3 --warmup 1 --no-cpu-baseline > /dev/null 2> $O/pmc_fetch_$w.err || exit 1
  timeout -s KILL 120 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/pmc_write_$w -- python3 bench.py --workload $w --ramp-ms 0 --steps 3 --warmup 1 --no-cpu-baseline > /dev/null 2> $O/pmc_write_$w.err || exit 1
done
python3 - <<PY
import csv,glob,collections,json
O='$O'
out={}
for w in ('tel1024x10s8k_f64_hann200_hop80_nfft256','music64x60s44k_f64_hann1024_hop256_nfft1024','music64x60s44k_f64_hann2048_hop512_nfft2048'):
    st=glob.glob(O+'/stats_'+w+'/**/*kernel_stats.csv',recursive=True)[0]
    rows=[r for r in csv.DictReader(open(st)) if 'stft' in r['Name']]
    res={'kernel':rows[0]['Name'],'calls':rows[0]['Calls'],'AverageNs':float(rows[0]['AverageNs'])}
    for kind in ('fetch','write'):
        f=glob.glob(O+'/pmc_'+kind+'_'+w+'/**/*counter_collection.csv',recursive=True)[0]
        v=[float(r['Counter_Value']) for r in csv.DictReader(open(f)) if 'stft' in r['Kernel_Name']]
        res[kind.upper()+'_SIZE_per_launch']=sum(v)/len(v)
    out[w]=res
json.dump(out,open(O+'/summary.json','w'),indent=1)
print(json.dumps(out,indent=1))
PY
