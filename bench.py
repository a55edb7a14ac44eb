#!/usr/bin/env python3
"""Benchmark of the fused STFT path: frames/s and achieved HBM GB/s on MI355X.

    python bench.py --gpus N --steps K --warmup W        (N > 1: launched by torch.distributed.run)

One step = one pass of the fused STFT kernel over one resident batch of synthetic signals (one
kernel launch).  Default workload: 1024 utterances x 10 s @ 16 kHz float32, Hann window 400, hop 160,
nfft 512 (1 024 000 frames per GPU per step); with N GPUs every rank holds its own 1024 utterances
(weak scaling, signals sharded by rank, no collective on the data path).  Inputs and outputs are
resident in HBM; host<->device copies are outside the timed region.

Prints ONE JSON line on rank 0.  `roofline.achieved` = algorithmic bytes per launch (every input
sample read once + window + every output value written once) / mean launch duration from HIP events
recorded on the launch stream.  `cpu_baseline` = the CPU oracle's T-precision build (a port of the
reference loop, one thread) timed on a stated sample of the same workload.
"""
import argparse
import ctypes as C
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBPS = 8000.0          # MI355X HBM3E spec peak

WORKLOADS = {
    # name: (dtype, B per GPU, N, window, L, hop, nfft, kind)
    "utt1024x10s_f32_hann400_hop160_nfft512": ("float32", 1024, 160000, "hann", 400, 160, 512, "stft"),
    "utt1x10s_f32_hann400_hop160_nfft512": ("float32", 1, 160000, "hann", 400, 160, 512, "stft"),
    "flt_f64_8ch_x_1h16k_povey400_hop160_nfft512": ("float64", 8, 57600000, "povey", 400, 160, 512, "flt"),
    "longwin_f32_8ch_x_1h48k_bh4096_hop1024_nfft4096": ("float32", 8, 172800000, "blackmanharris", 4096, 1024, 4096, "stft"),
}
DEFAULT = "utt1024x10s_f32_hann400_hop160_nfft512"


def cpu_baseline(x_host, win, stp, nfft, kind, W, budget_s=12.0):
    """Times the oracle's T-precision build (oracle/liborc_fast.so, -O3 -ffast-math, one thread) on as
    many whole signals of the batch as fit the budget; returns (frames/s, description)."""
    path = os.path.join(ROOT, "oracle", "liborc_fast.so")
    lib = C.CDLL(path)
    sfx = "s" if x_host.dtype == np.float32 else "d"
    ct = C.c_float if sfx == "s" else C.c_double
    N, L = x_host.shape[1], win.size
    F = nfft // 2 + 1
    Y = np.empty((W, F), dtype=x_host.dtype)
    vp = lambda a: a.ctypes.data_as(C.c_void_p)  # noqa: E731
    sz = C.c_size_t

    def one(b):
        if kind == "flt":
            f = getattr(lib, f"orc_stft_flt_{sfx}")
            f.restype = C.c_int
            return f(vp(Y), vp(x_host[b]), vp(win), sz(N), sz(L), sz(W), sz(nfft), ct(L // 2), ct(stp), 0, 0, 0, 1, 1)
        f = getattr(lib, f"orc_stft_{sfx}")
        f.restype = C.c_int
        return f(vp(Y), vp(x_host[b]), vp(win), sz(N), sz(L), sz(nfft), sz(stp), 0, 0, 0, 0, 1)

    assert one(0) == 0          # warm-up (twiddle table, page faults)
    t0 = time.perf_counter()
    done = 0
    nb = x_host.shape[0]
    while time.perf_counter() - t0 < budget_s:      # whole signals, cycling through the sample
        assert one(done % nb) == 0
        done += 1
    dt = time.perf_counter() - t0
    return done * W / dt, (f"{done} signal transforms cycling over {nb} signals of the batch "
                           f"({done * W} frames, {dt:.1f} s, one thread)")


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--workload", default=DEFAULT, choices=sorted(WORKLOADS))
    ap.add_argument("--ramp-ms", type=float, default=300.0, help="untimed device work before the warmup steps (clock ramp); 0 disables")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--gather", action="store_true", help="also time one RCCL all-gather of the output shards")
    ap.add_argument("--generic", action="store_true", help="force the generic kernel family (A/B)")
    args = ap.parse_args()

    import torch
    import torch.distributed as dist

    import dsp_amd

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run --nproc-per-node {args.gpus}")
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if world > 1:
        dist.init_process_group("nccl", device_id=dev)

    dtname, B, N, wname, L, stp, nfft, kind = WORKLOADS[args.workload]
    tdt = torch.float32 if dtname == "float32" else torch.float64
    ndt = np.float32 if dtname == "float32" else np.float64
    esz = np.dtype(ndt).itemsize
    F = nfft // 2 + 1
    win_h = dsp_amd.WINDOWS[wname](L, ndt)
    win = torch.from_numpy(win_h).to(dev)
    # synthetic signals generated on the owning GPU; seed depends on the global signal block
    g = torch.Generator(device=dev).manual_seed(0xD5F0 + rank)
    x = torch.empty((B, N), dtype=tdt, device=dev)
    for b0 in range(0, B, 64):
        x[b0:b0 + 64] = torch.randn((min(64, B - b0), N), generator=g, device=dev, dtype=tdt) * 0.1
    if kind == "flt":
        W = (N - 1 - L // 2) // stp
        dsp_amd.set_flt_mode(False)
        run = lambda out: dsp_amd.stft_flt_dev(x, win, W, float(L // 2), float(stp), out=out)  # noqa: E731
    else:
        W = dsp_amd.num_frames(N, L, stp, False)
        run = lambda out: dsp_amd.stft_dev(x, win, stp, out=out)  # noqa: E731
    if args.generic:
        dsp_amd.force_generic(True)
    y = torch.empty((B, W, F), dtype=tdt, device=dev)

    def sync_all():
        torch.cuda.synchronize(dev)
        if world > 1:
            dist.barrier()
            torch.cuda.synchronize(dev)

    # A GPU that has been idle takes some tens of milliseconds of work to reach its sustained clocks (the
    # first ~50 launches of this workload run 8-10 % slower): bring it there before the W warmup steps, so
    # that the K timed steps measure the steady state whatever K and W are.
    if args.ramp_ms > 0:
        t_r = time.perf_counter()
        while (time.perf_counter() - t_r) * 1e3 < args.ramp_ms:
            for _ in range(20):
                run(y)
            torch.cuda.synchronize(dev)
    for _ in range(args.warmup):
        run(y)
    sync_all()
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(args.steps)]
    t0 = time.perf_counter()
    for a, b in ev:
        a.record()
        run(y)
        b.record()
    sync_all()
    elapsed = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    kern_ms = float(np.mean([a.elapsed_time(b) for a, b in ev]))

    gather_ms = None
    if args.gather and world > 1:
        full = torch.empty((world * B, W, F), dtype=tdt, device=dev)
        dist.all_gather_into_tensor(full, y)
        sync_all()
        t1 = time.perf_counter()
        dist.all_gather_into_tensor(full, y)
        sync_all()
        gather_ms = (time.perf_counter() - t1) * 1e3
        del full

    frames_per_step = world * B * W
    value = frames_per_step * args.steps / elapsed
    alg_bytes = B * N * esz + L * esz + B * W * F * esz           # per launch (one GPU)
    achieved = alg_bytes / (kern_ms * 1e-3) / 1e9

    traffic = None
    tpath = os.path.join(ROOT, "profiles", "traffic.json")
    if os.path.exists(tpath):
        traffic = json.load(open(tpath)).get(args.workload)

    out = {
        "metric": "stft_frames_per_sec", "value": value, "unit": "frames/s", "n_gpus": world, "steps": args.steps,
        "warmup": args.warmup, "ms_per_step": elapsed / args.steps * 1e3, "higher_is_better": True, "scaling": "weak",
        "vs_baseline": None, "dtype": "f32" if esz == 4 else "f64", "data": "synthetic",
        "config": {"workload": args.workload, "signals_per_gpu": B, "samples_per_signal": N, "window": wname, "L": L,
                   "hop": stp, "nfft": nfft, "frames_per_signal": W, "entry_point": "dsp_stft_flt_batch_dev" if kind == "flt" else "dsp_stft_batch_dev",
                   "kernel_family": "generic" if args.generic else {1: "direct", 2: "pow2", 3: "r512", 4: "r4096", 5: "wave"}[dsp_amd.kernel_family(nfft, ndt)],
                   "sharding": f"{world} ranks x {B} signals, no data-path collective", "device_ramp_ms": args.ramp_ms},
        "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBPS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBPS,
                     "traffic": traffic, "algorithmic_bytes_per_launch": alg_bytes, "kernel_ms": kern_ms,
                     "nominal_tflops_5NlogN": 5 * nfft * np.log2(nfft) * B * W / (kern_ms * 1e-3) / 1e12},
    }
    if gather_ms is not None:
        out["gather_ms"] = gather_ms
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        nb = min(B, 256) if N <= 1_000_000 else 1
        x_host = x[:nb].cpu().numpy() if N <= 1_000_000 else x[:1, : 16000 * 60].cpu().numpy()
        Wc = W if N <= 1_000_000 else ((x_host.shape[1] - 1 - L // 2) // stp if kind == "flt" else dsp_amd.num_frames(x_host.shape[1], L, stp, False))
        v, desc = cpu_baseline(x_host, win_h, stp, nfft, kind, Wc)
        out["cpu_baseline"] = {"value": v, "unit": "frames/s", "cores": 1, "kind": "port", "sample": desc}
    if rank == 0:
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
