"""Copies what tools/experiments/prof_f64.sh left under gpurun_out/prof_f64/ into profiles/ and adds the three float64
workloads to profiles/traffic.json (HBM bytes per launch = 2 * FETCH_SIZE KiB + WRITE_SIZE KiB, as for the others)."""
import glob, json, shutil, sys
tag = sys.argv[1] if len(sys.argv) > 1 else 'r02'
src = 'gpurun_out/prof_f64'
summ = json.load(open(f'{src}/summary.json'))
traffic = json.load(open('profiles/traffic.json'))
for w, v in summ.items():
    st = glob.glob(f'{src}/stats_{w}/**/*kernel_stats.csv', recursive=True)[0]
    shutil.copy(st, f'profiles/{tag}_kernel_stats_{w}.csv')
    shutil.copy(f'{src}/bench_under_rocprof_{w}.json', f'profiles/{tag}_bench_under_rocprof_{w}.json')
    f, wr = v['FETCH_SIZE_per_launch'] * 2048.0, v['WRITE_SIZE_per_launch'] * 1024.0
    traffic[w] = f + wr
    traffic['_fetch_bytes_' + w] = f
    traffic['_write_bytes_' + w] = wr
    print(w, v['AverageNs'], 'traffic', traffic[w])
json.dump(summ, open(f'profiles/{tag}_rocprofv3_summary_f64.json', 'w'), indent=1)
json.dump(traffic, open('profiles/traffic.json', 'w'), indent=1)
