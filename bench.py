#!/usr/bin/env python3
"""Benchmark of the fused STFT path: frames/s and achieved HBM GB/s on MI355X.

    python bench.py --gpus N --steps K --warmup W        (N > 1: launched by torch.distributed.run)

One step = one pass of the fused STFT kernel over one resident batch of synthetic signals (one kernel
launch).  N = 1: 1024 utterances x 10 s @ 16 kHz float32, Hann window 400, hop 160, nfft 512 (1 024 000
frames per step).  N > 1: every rank holds 8192 such utterances (at N = 8 the 65 536 utterances of the
sharded configuration), signals sharded by rank, no collective on the data path; the gather of the
spectrogram shards to rank 0 (RCCL) is timed separately, alone and overlapped with the computation in
pieces, and is never part of `value`.  Inputs and outputs are resident in HBM; host<->device copies are
outside the timed region.

Prints ONE JSON line on rank 0.  `roofline.achieved` = algorithmic bytes per launch (every input sample
read once + window + every output value written once) / mean launch duration from HIP events recorded on
the launch stream.  `cpu_baseline` = the CPU oracle's T-precision build (a port of the reference loop; on
FFTW itself where the host has libfftw3) timed on a stated sample of the same workload, one thread (the
reference is single-threaded) and on all the cores this process may use.
"""
import argparse
import ctypes as C
import json
import os
import subprocess
import sys
import tempfile
import threading
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBPS = 8000.0          # MI355X HBM3E spec peak
FP32_VALU_PEAK_TFLOPS = 157.3   # MI355X FP32 vector peak: 256 CUs x 4 SIMDs x 64 flops per cycle x 2.4 GHz
# what a loop of nothing but packed fmas sustains (tools/experiments/ubench/clock_cal.hip: the shader clock settles at
# 1.975 GHz under that load, 4 cycles per packed wave instruction)
FP32_VALU_SUSTAINED_TFLOPS = 129.4

WORKLOADS = {
    # name: (dtype, B per GPU, N, window, L, hop, nfft, kind)
    "utt1024x10s_f32_hann400_hop160_nfft512": ("float32", 1024, 160000, "hann", 400, 160, 512, "stft"),
    "utt8192x10s_f32_hann400_hop160_nfft512": ("float32", 8192, 160000, "hann", 400, 160, 512, "stft"),
    "utt1x10s_f32_hann400_hop160_nfft512": ("float32", 1, 160000, "hann", 400, 160, 512, "stft"),
    "flt_f64_8ch_x_1h16k_povey400_hop160_nfft512": ("float64", 8, 57600000, "povey", 400, 160, 512, "flt"),
    "longwin_f32_8ch_x_1h48k_bh4096_hop1024_nfft4096": ("float32", 8, 172800000, "blackmanharris", 4096, 1024, 4096, "stft"),
    # the 64-channel forms of the long-signal workloads as single launches: all 64 float64 channels (29.5 GB in, 47.4 GB out) and
    # 32 of the 128 float32 channels (22.1 GB in, 44.3 GB out; all 128 are 265 GB and do not fit one GPU beside anything else)
    "flt_f64_64ch_x_1h16k_povey400_hop160_nfft512": ("float64", 64, 57600000, "povey", 400, 160, 512, "flt"),
    "longwin_f32_32ch_x_1h48k_bh4096_hop1024_nfft4096": ("float32", 32, 172800000, "blackmanharris", 4096, 1024, 4096, "stft"),
    # the other transform lengths with kernels of their own (not headline configurations)
    "tel1024x10s8k_f32_hann200_hop80_nfft256": ("float32", 1024, 80000, "hann", 200, 80, 256, "stft"),
    "music64x60s44k_f32_hann1024_hop256_nfft1024": ("float32", 64, 2646000, "hann", 1024, 256, 1024, "stft"),
    "music64x60s44k_f32_hann2048_hop512_nfft2048": ("float32", 64, 2646000, "hann", 2048, 512, 2048, "stft"),
    "tel1024x10s8k_f64_hann200_hop80_nfft256": ("float64", 1024, 80000, "hann", 200, 80, 256, "stft"),
    "music64x60s44k_f64_hann1024_hop256_nfft1024": ("float64", 64, 2646000, "hann", 1024, 256, 1024, "stft"),
    "music64x60s44k_f64_hann2048_hop512_nfft2048": ("float64", 64, 2646000, "hann", 2048, 512, 2048, "stft"),
}
DEFAULT_1GPU = "utt1024x10s_f32_hann400_hop160_nfft512"
DEFAULT_NGPU = "utt8192x10s_f32_hann400_hop160_nfft512"


def host_cores():
    """Cores this process may run on (the box's share, not the machine's count), capped by OMP_NUM_THREADS."""
    try:
        n = len(os.sched_getaffinity(0))
    except AttributeError:
        n = os.cpu_count() or 1
    cap = os.environ.get("OMP_NUM_THREADS")
    if cap and cap.isdigit() and int(cap) > 0:
        n = min(n, int(cap))
    return max(1, n)


def find_fftw(lib, sfx):
    """Hands FFTW's planner to the oracle if libfftw3(f).so.3 can be loaded on this host; returns a label."""
    name = "libfftw3f.so.3" if sfx == "s" else "libfftw3.so.3"
    pre = "fftwf_" if sfx == "s" else "fftw_"
    try:
        f = C.CDLL(name)
        ptr = lambda n: C.cast(getattr(f, pre + n), C.c_void_p)  # noqa: E731
        getattr(lib, f"orc_set_fftw_{sfx}")(ptr("plan_r2r_1d"), ptr("execute"), ptr("destroy_plan"))
        return True, f"{name} found: reference loop on FFTW (r2r R2HC plan, FFTW_ESTIMATE, one plan per call)"
    except (OSError, AttributeError) as e:
        return False, f"{name} not loadable on this host ({type(e).__name__}): in-repo port of the reference loop"


def cpu_baseline(x_host, win, stp, nfft, kind, W, budget_s=10.0):
    """Times the oracle's T-precision build (oracle/liborc_fast.so, -O3 -ffast-math) on whole signals of the
    sample for `budget_s` seconds: one thread, then one thread per core.  Returns the cpu_baseline object."""
    lib = C.CDLL(os.path.join(ROOT, "oracle", "liborc_fast.so"))
    sfx = "s" if x_host.dtype == np.float32 else "d"
    ct = C.c_float if sfx == "s" else C.c_double
    N, L = x_host.shape[1], win.size
    F = nfft // 2 + 1
    vp = lambda a: a.ctypes.data_as(C.c_void_p)  # noqa: E731
    sz = C.c_size_t
    have_fftw, fftw_note = find_fftw(lib, sfx)
    mode = 2 if have_fftw else 1
    f_flt, f_stft = getattr(lib, f"orc_stft_flt_{sfx}"), getattr(lib, f"orc_stft_{sfx}")
    f_flt.restype = f_stft.restype = C.c_int
    nb = x_host.shape[0]

    def one(Y, b):
        if kind == "flt":
            return f_flt(vp(Y), vp(x_host[b]), vp(win), sz(N), sz(L), sz(W), sz(nfft), ct(L // 2), ct(stp), 0, 0, 0, 1, mode)
        return f_stft(vp(Y), vp(x_host[b]), vp(win), sz(N), sz(L), sz(nfft), sz(stp), 0, 0, 0, 0, mode)

    def leg(threads):
        done = [0] * threads
        t0 = time.perf_counter()

        def work(i):
            Y = np.empty((W, F), dtype=x_host.dtype)
            assert one(Y, i % nb) == 0                  # warm-up (twiddle table, page faults)
            k = i
            while time.perf_counter() - t0 < budget_s:  # whole signals, cycling through the sample
                assert one(Y, k % nb) == 0              # (the foreign call releases the interpreter lock)
                k += threads
                done[i] += 1
        ts = [threading.Thread(target=work, args=(i,)) for i in range(threads)]
        for t in ts:
            t.start()
        for t in ts:
            t.join()
        dt = time.perf_counter() - t0
        return sum(done) * W / dt, sum(done), dt

    def reference_chain(budget):
        """The same spectrogram by the reference's OWN code where that builds without FFTW (oracle/_ref/libdspref.so,
        compiled from the reference's sources where they lie): c/window_univar(_flt).c -> c/fft.rad2.c (complex entry
        point, rows padded to nfft) -> c/fft_power.c, one thread.  Not what its stft runs (that is FFTW's r2c), so it is
        reported beside the port, not instead of it."""
        try:
            from oracle import oracle as O
            if O.ref() is None:
                return None
            t0 = time.perf_counter()
            n = 0
            while n == 0 or time.perf_counter() - t0 < budget:
                xs = x_host[n % nb]
                if kind == "flt":
                    rc, fr = O.ref_window_univar_flt(xs, win, W, float(L // 2), float(stp))
                else:
                    rc, fr = O.ref_window_univar(xs, win, stp, False)
                assert rc == 0
                for lo in range(0, fr.shape[0], 4096):          # bounded temporaries
                    rc, sp = O.ref_fft_rad2(fr[lo:lo + 4096], nfft)
                    assert rc == 0
                    rc, _ = O.ref_fft_power(sp)
                    assert rc == 0
                n += 1
            dt = time.perf_counter() - t0
            return {"value": n * W / dt, "unit": "frames/s", "cores": 1, "kind": "reference",
                    "sample": f"{n} signal(s), {n * W} frames, {dt:.1f} s: window_univar -> fft_rad2 (complex, padded rows) -> fft_power from the reference's sources; the padding copy is numpy's"}
        except Exception as e:  # the checker's helper library is optional on the box
            return {"error": repr(e)[:200]}

    cores = host_cores()
    v1, n1, t1 = leg(1)
    vall, nall, tall = leg(cores) if cores > 1 else (v1, n1, t1)
    return {
        "value": v1, "unit": "frames/s", "cores": 1, "kind": "port",
        "sample": (f"{n1} signal transforms cycling over {nb} signal(s) of the batch, {x_host.shape[1]} samples each "
                   f"({n1 * W} frames, {t1:.1f} s, one thread: the reference is single-threaded, c/fft.fftw.r2hc.c:8)"),
        "all_cores": {"value": vall, "unit": "frames/s", "cores": cores,
                      "sample": f"{nall} signal transforms, one signal per thread at a time ({nall * W} frames, {tall:.1f} s)"},
        "host": {"cores_usable": cores, "cpu_count": os.cpu_count(), "fft": fftw_note},
        "reference_chain": reference_chain(min(5.0, budget_s)),
    }


def config1_extras(dsp_amd, x, win_h, stp, nfft):
    """One utterance through the drop-in boundary: the host-pointer stft_s (copies included) and the stft
    command-line tool as a process (HIP set-up, code object, file I/O included), beside the device-resident call."""
    import torch
    xh = x[0].cpu().numpy()
    for _ in range(20):
        dsp_amd.stft(xh, win_h, stp)
    t0 = time.perf_counter()
    for _ in range(200):
        dsp_amd.stft(xh, win_h, stp)
    host_us = (time.perf_counter() - t0) / 200 * 1e6
    win = torch.from_numpy(win_h).to(x.device)
    y = dsp_amd.stft_dev(x, win, stp)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(200):
        dsp_amd.stft_dev(x, win, stp, out=y)
        torch.cuda.synchronize()
    dev_us = (time.perf_counter() - t0) / 200 * 1e6
    out = {"stft_s_host_pointers_us_per_call": host_us, "stft_batch_dev_plus_sync_us_per_call": dev_us}
    # a serving loop of single utterances is launch-bound: 100 calls of the allocation-free entry point captured into one
    # HIP graph and replayed, against the same 100 calls enqueued one by one (both without a sync in between)
    try:
        import ctypes as C
        from dsp_amd._lib import lib

        def na_call():
            return lib().dsp_stft_batch_dev_noalloc_s(C.c_void_p(y.data_ptr()), C.c_void_p(x.data_ptr()), C.c_void_p(win.data_ptr()), 1, x.shape[1], x.shape[1],
                                                      win.numel(), nfft, stp, 0, 0, 0, 0, C.c_void_p(torch.cuda.current_stream().cuda_stream))
        assert na_call() == 0
        torch.cuda.synchronize()
        graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(graph):
            for _ in range(100):
                assert na_call() == 0
        for fn, key in ((graph.replay, "graph_of_100_calls_us_per_call"), (lambda: [na_call() for _ in range(100)], "enqueued_100_calls_us_per_call")):
            fn()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(20):
                fn()
            torch.cuda.synchronize()
            out[key] = (time.perf_counter() - t0) / 20 / 100 * 1e6
    except Exception as e:  # noqa: BLE001
        out["graph_of_100_calls_us_per_call"] = f"{type(e).__name__}: {e}"
    # the kernel alone, warm (its 1.7 MB working set never leaves the caches) and cold (512 MB written between launches)
    scrub = torch.empty(128 << 20, dtype=torch.float32, device=x.device)
    for name, cold in (("kernel_us_warm", False), ("kernel_us_cold", True)):
        ts = []
        for i in range(30):
            if cold:
                scrub.fill_(float(i))
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record()
            dsp_amd.stft_dev(x, win, stp, out=y)
            b.record()
            torch.cuda.synchronize()
            ts.append(a.elapsed_time(b) * 1e3)
        out[name] = {"median": float(np.median(ts)), "min": float(np.min(ts))}
    del scrub
    tool = os.path.join(ROOT, "bin", "stft")
    if os.path.exists(tool):
        with tempfile.TemporaryDirectory() as d:
            np.save(os.path.join(d, "x.npy"), xh)
            np.save(os.path.join(d, "w.npy"), win_h)
            walls = []
            for _ in range(5):
                t0 = time.perf_counter()
                rc = subprocess.run([tool, "-s", str(stp), "-o", os.path.join(d, "y.npy"), os.path.join(d, "x.npy"), os.path.join(d, "w.npy")]).returncode
                walls.append((time.perf_counter() - t0) * 1e3)
                if rc != 0:
                    break
            if rc == 0:
                out["stft_tool_process_wall_ms"] = {"min": min(walls), "median": float(np.median(walls)), "runs": len(walls)}
    return out


def gather_leg(x, win, B, world, rank, compute, chunks, sync, **kw):
    """The gather of the shards to rank 0, as dsp_amd.dist does it: (i) alone, after the shard is computed;
    (ii) computation and gather pipelined in `chunks` pieces.  Returns (timings in ms, Y_full on rank 0)."""
    from dsp_amd.dist import stft_sharded
    res = {}
    sync()
    t0 = time.perf_counter()
    y, full = stft_sharded(x, win, world * B, rank, world, gather="root", compute=compute, chunks=1, **kw)
    sync()
    res["compute_then_gather_ms"] = (time.perf_counter() - t0) * 1e3
    del full, y
    sync()
    t0 = time.perf_counter()
    y, full = stft_sharded(x, win, world * B, rank, world, gather="root", compute=compute, chunks=chunks, **kw)
    sync()
    res["pipelined_ms"] = (time.perf_counter() - t0) * 1e3
    res["chunks"] = chunks
    return res, full


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--workload", default=None, choices=sorted(WORKLOADS))
    ap.add_argument("--ramp-ms", type=float, default=300.0, help="untimed device work before the warmup steps (clock ramp); 0 disables")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-gather", action="store_true", help="N > 1: skip the (separately timed) gather of the shards to rank 0")
    ap.add_argument("--gather-chunks", type=int, default=8)
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
    # DSP_BENCH_REHEARSAL=1: all ranks on GPU 0 with the gloo backend, to rehearse the N > 1 path on a one-GPU box
    rehearsal = os.environ.get("DSP_BENCH_REHEARSAL") == "1"
    if rehearsal:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if world > 1:
        if rehearsal:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=dev)
    workload = args.workload or (DEFAULT_1GPU if world == 1 else DEFAULT_NGPU)

    dtname, B, N, wname, L, stp, nfft, kind = WORKLOADS[workload]
    tdt = torch.float32 if dtname == "float32" else torch.float64
    ndt = np.float32 if dtname == "float32" else np.float64
    esz = np.dtype(ndt).itemsize
    F = nfft // 2 + 1
    win_h = dsp_amd.WINDOWS[wname](L, ndt)
    win = torch.from_numpy(win_h).to(dev)
    # synthetic signals generated on the owning GPU, a seed per block of 64 signals of the WHOLE job: a signal's samples
    # do not depend on the number of GPUs the job is cut over
    g = torch.Generator(device=dev)
    x = torch.empty((B, N), dtype=tdt, device=dev)
    for b0 in range(0, B, 64):
        g.manual_seed(0xD5F0 + (rank * B + b0) // 64)
        for b1 in range(b0, min(B, b0 + 64), 8):          # (8 at a time: the generator's stream is the same, the temporaries stay small)
            n8 = min(8, B - b1, b0 + 64 - b1)
            x[b1:b1 + n8] = torch.randn((n8, N), generator=g, device=dev, dtype=tdt)
            x[b1:b1 + n8] *= 0.1
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
    kern_all = [a.elapsed_time(b) for a, b in ev]
    kern_ms = float(np.mean(kern_all))

    gather = None
    if world > 1 and not args.no_gather and kind == "stft":
        # separately timed, never part of `value`; a failure here must not lose the throughput line
        try:
            del y
            res, full = gather_leg(x, win, B, world, rank, dsp_amd.stft_dev, args.gather_chunks, sync_all, stp=stp)
            tm = torch.tensor([res["compute_then_gather_ms"], res["pipelined_ms"]], dtype=torch.float64, device=dev)
            dist.all_reduce(tm, op=dist.ReduceOp.MAX)
            gather = {"to": "rank 0", "backend": dist.get_backend(), "rccl_world_size": dist.get_world_size(),
                      "bytes_per_rank": B * W * F * esz, "compute_then_gather_ms": float(tm[0].item()),
                      "pipelined_ms": float(tm[1].item()), "chunks": res["chunks"]}
            del full
        except Exception as e:  # noqa: BLE001
            gather = {"error": f"{type(e).__name__}: {e}"}

    frames_per_step = world * B * W
    value = frames_per_step * args.steps / elapsed
    alg_bytes = B * N * esz + L * esz + B * W * F * esz           # per launch (one GPU)
    achieved = alg_bytes / (kern_ms * 1e-3) / 1e9
    tflops = 5 * nfft * np.log2(nfft) * B * W / (kern_ms * 1e-3) / 1e12

    # HBM bytes per launch from the PMC counters: collected by separate rocprofv3 passes (tools/experiments/prof_round.sh),
    # not in this run
    traffic = None
    tpath = os.path.join(ROOT, "profiles", "traffic.json")
    if os.path.exists(tpath):
        traffic = json.load(open(tpath)).get(workload)

    out = {
        "metric": "stft_frames_per_sec", "value": value, "unit": "frames/s", "n_gpus": world, "steps": args.steps,
        "warmup": args.warmup, "ms_per_step": elapsed / args.steps * 1e3, "higher_is_better": True, "scaling": "weak",
        "vs_baseline": None, "dtype": "f32" if esz == 4 else "f64", "data": "synthetic",
        "config": {"workload": workload, "signals_per_gpu": B, "samples_per_signal": N, "window": wname, "L": L,
                   "hop": stp, "nfft": nfft, "frames_per_signal": W, "entry_point": "dsp_stft_flt_batch_dev" if kind == "flt" else "dsp_stft_batch_dev",
                   "kernel_family": "generic" if args.generic else {1: "direct", 2: "pow2", 3: "r512", 4: "r4096", 5: "wave", 6: "r512x", 7: "r256"}[dsp_amd.kernel_family(nfft, ndt)],
                   "sharding": f"{world} ranks x {B} signals, no data-path collective", "device_ramp_ms": args.ramp_ms},
        "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBPS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBPS,
                     "traffic": traffic, "traffic_source": "profiles/traffic.json (separate rocprofv3 --pmc passes)" if traffic else None,
                     "algorithmic_bytes_per_launch": alg_bytes, "kernel_ms": kern_ms, "kernel_ms_median": float(np.median(kern_all)), "kernel_ms_min": float(np.min(kern_all)),
                     "nominal_tflops_5NlogN": tflops},
    }
    if esz == 4:
        # the other bound of this path: float32 vector issue, on the nominal 5 N log2 N flops (nfft 4096 sits on the ridge)
        out["roofline_valu"] = {"bound": "fp32_valu", "achieved": tflops, "peak": FP32_VALU_PEAK_TFLOPS, "unit": "TFLOP/s",
                                "frac": tflops / FP32_VALU_PEAK_TFLOPS, "sustained_peak": FP32_VALU_SUSTAINED_TFLOPS,
                                "sustained_frac": tflops / FP32_VALU_SUSTAINED_TFLOPS}
    if gather is not None:
        out["gather"] = gather
    if rank == 0 and world == 1 and B == 1:
        out["one_utterance"] = config1_extras(dsp_amd, x, win_h, stp, nfft)
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        nb = min(B, 256) if N <= 1_000_000 else 1
        x_host = x[:nb].cpu().numpy() if N <= 1_000_000 else x[:1, : 16000 * 60].cpu().numpy()
        Wc = W if N <= 1_000_000 else ((x_host.shape[1] - 1 - L // 2) // stp if kind == "flt" else dsp_amd.num_frames(x_host.shape[1], L, stp, False))
        out["cpu_baseline"] = cpu_baseline(x_host, win_h, stp, nfft, kind, Wc)
    if rank == 0:
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
