"""Shader clock and board power while one workload's kernel runs back to back for several seconds.

    python tools/experiments/clock_telemetry.py <workload of bench.py> [seconds]

A thread reads the driver's own files (sysfs: current sclk / mclk level, hwmon power and frequency) and, if they are
on the box, `rocm-smi` / `amd-smi`, while the main thread keeps the launch queue full.  Prints one JSON object."""
import glob, json, os, re, subprocess, sys, threading, time
sys.path.insert(0, '.')
import numpy as np, torch, dsp_amd
import bench

name = sys.argv[1]
secs = float(sys.argv[2]) if len(sys.argv) > 2 else 6.0
dtype, B, N, wname, L, hop, nfft, kind = bench.WORKLOADS[name]
dt = np.float32 if dtype == 'float32' else np.float64
x = (torch.randn((B, N), device='cuda', dtype=torch.float32 if dt == np.float32 else torch.float64) * 0.1)
win = torch.from_numpy(getattr(dsp_amd, wname)(L, dt)).cuda()
if kind == 'flt':
    W = int((N - 1 - L // 2) // hop)
    run = lambda out=None: dsp_amd.stft_flt_dev(x, win, W, float(L // 2), float(hop), nfft=nfft, out=out)
else:
    run = lambda out=None: dsp_amd.stft_dev(x, win, hop, nfft=nfft, out=out)
y = run()
torch.cuda.synchronize()

def rd(p):
    try:
        with open(p) as f:
            return f.read().strip()
    except OSError:
        return None

cards = [c for c in sorted(glob.glob('/sys/class/drm/card[0-9]*/device')) if rd(c + '/pp_dpm_sclk') is not None]
hw = sorted(glob.glob('/sys/class/drm/card[0-9]*/device/hwmon/hwmon*'))
info = {'sysfs_cards': cards, 'hwmon': hw}
samples = []
stop = False

def smi(cmd):
    try:
        return subprocess.run(cmd, capture_output=True, text=True, timeout=10).stdout
    except Exception as e:           # noqa: BLE001
        return 'ERR ' + repr(e)

def sample():
    s = {'t': time.perf_counter()}
    for c in cards:
        cur = [l for l in (rd(c + '/pp_dpm_sclk') or '').splitlines() if l.endswith('*')]
        s['sclk ' + c.split('/')[4]] = cur[0] if cur else None
        cur = [l for l in (rd(c + '/pp_dpm_mclk') or '').splitlines() if l.endswith('*')]
        s['mclk ' + c.split('/')[4]] = cur[0] if cur else None
        s['busy ' + c.split('/')[4]] = rd(c + '/gpu_busy_percent')
    for h in hw:
        for f in ('power1_average', 'power1_input', 'freq1_input', 'freq2_input', 'temp1_input'):
            v = rd(h + '/' + f)
            if v is not None:
                s[f + ' ' + h.split('/')[4]] = v
    return s

def sampler():
    while not stop:
        samples.append(sample())
        time.sleep(0.05)

info['idle'] = sample()
info['rocm_smi_idle'] = smi(['rocm-smi', '--showclocks', '--showpower'])[-1500:]
th = threading.Thread(target=sampler)
th.start()
t0 = time.perf_counter()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
n = 0
smi_busy = None
e0.record()
while time.perf_counter() - t0 < secs:
    for _ in range(50):
        run(out=y)
    n += 50
    torch.cuda.synchronize()
    if smi_busy is None and time.perf_counter() - t0 > secs / 2:
        # one reading of the tool itself, taken with launches queued behind it
        for _ in range(400):
            run(out=y)
        n += 400
        smi_busy = smi(['rocm-smi', '--showclocks', '--showpower'])[-1500:]
        torch.cuda.synchronize()
e1.record()
torch.cuda.synchronize()
stop = True
th.join()
info['rocm_smi_busy'] = smi_busy
info['launches'] = n
info['ms_per_launch_incl_gaps'] = e0.elapsed_time(e1) / n
keys = sorted({k for s in samples for k in s if k != 't'})
summ = {}
for k in keys:
    vals = [s[k] for s in samples[len(samples) // 4:] if s.get(k) is not None]     # the first quarter is ramp
    nums = []
    for v in vals:
        m = re.search(r'(\d+)\s*Mhz', v, re.I) or re.fullmatch(r'(\d+)', v)
        if m:
            nums.append(float(m.group(1)))
    if nums:
        summ[k] = {'n': len(nums), 'mean': float(np.mean(nums)), 'min': min(nums), 'max': max(nums)}
    elif vals:
        summ[k] = {'n': len(vals), 'last': vals[-1]}
info['busy_summary'] = summ
info['workload'] = name
print(json.dumps(info, indent=1))
