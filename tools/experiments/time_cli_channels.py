"""64 channels through the stft tool: one process with -d against one process per channel (what the tool could do before).
    python tools/experiments/time_cli_channels.py            (on the GPU box, from the repository root)"""
import os, subprocess, sys, time, tempfile
import numpy as np
sys.path.insert(0, '.')
C_, N = 64, 960000                       # 64 channels x 1 min @ 16 kHz float32
rng = np.random.default_rng(3)
sig = (rng.standard_normal((C_, N)) * 0.1).astype(np.float32)
win = (0.5 - 0.5 * np.cos(2 * np.pi * np.arange(400) / 399)).astype(np.float32)
d = tempfile.mkdtemp(dir='/tmp')
np.save(f'{d}/w.npy', win)
np.save(f'{d}/rows.npy', sig)                             # [C][N]
np.save(f'{d}/inter.npy', np.ascontiguousarray(sig.T))    # [N][C]
for c in range(C_):
    np.save(f'{d}/ch{c}.npy', sig[c])
def t(cmd):
    t0 = time.perf_counter(); r = subprocess.run(cmd, capture_output=True); dt = time.perf_counter() - t0
    assert r.returncode == 0, r.stderr
    return dt
t([f'bin/stft', f'{d}/ch0.npy', f'{d}/w.npy', '-o', f'{d}/y.npy'])     # page the tool in
one = min(t(['bin/stft', '-d1', f'{d}/rows.npy', f'{d}/w.npy', '-o', f'{d}/y_rows.npy']) for _ in range(3))
inter = min(t(['bin/stft', '-d0', f'{d}/inter.npy', f'{d}/w.npy', '-o', f'{d}/y_inter.npy']) for _ in range(3))
t0 = time.perf_counter()
for c in range(C_):
    t(['bin/stft', f'{d}/ch{c}.npy', f'{d}/w.npy', '-o', f'{d}/y{c}.npy'])
per = time.perf_counter() - t0
Y = np.load(f'{d}/y_rows.npy'); Yi = np.load(f'{d}/y_inter.npy')
same = all(np.array_equal(Y[c], np.load(f'{d}/y{c}.npy')) for c in range(C_)) and np.array_equal(Y, Yi)
print(f'64 channels x 60 s @ 16 kHz float32 ({sig.nbytes / 1e6:.0f} MB in, {Y.nbytes / 1e6:.0f} MB out): one process -d1 (contiguous) {one * 1e3:.0f} ms, '
      f'one process -d0 (interleaved) {inter * 1e3:.0f} ms, 64 processes {per * 1e3:.0f} ms ({per / C_ * 1e3:.0f} ms each); same bits: {same}')
