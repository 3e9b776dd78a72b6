#!/usr/bin/env python3
"""Headline benchmark: complex64 IQ Msamples/s through Hann-windowed 4096-point
FFT + log-PSD (BASELINE.json configs[1]: 2^20 synthetic frames per GPU), with the
dominant kernel's achieved HBM GB/s against the 8 TB/s roofline and the numpy
oracle timed on the host cores of the same box.

    python bench.py --gpus N --steps K --warmup W

N > 1 is launched by the driver as
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 ...
one rank per GPU; rank g owns frames [g*2^20, (g+1)*2^20) of BASELINE.json config 4
(frame-range sharding, weak scaling, no data-path collective: the only RCCL traffic
is the barrier and the max-reduce of the elapsed time).

A "step" is one pass of the hot path over the rank's device-resident batch.  Rank 0
prints ONE JSON line.
"""
from __future__ import annotations

import argparse
import ctypes
import json
import os
import sys
import time

REPO = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, REPO)

NFFT = 4096
ALGO_BYTES_PER_SAMPLE = 12          # 8 B complex64 read + 4 B float32 written
HBM_PEAK_GBPS = 8000.0              # MI355X HBM3E spec peak (MI355X_MICROARCH.md)
METRIC = "complex64 IQ Msamples/s through windowed FFT+log-PSD, N=4096; % HBM3E peak"


def parse_args():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--frames", type=int, default=1 << 20, help="frames per GPU (default 2^20)")
    ap.add_argument("--window", default="hann", choices=["hann", "rect"])
    ap.add_argument("--cpu-seconds", type=float, default=10.0,
                    help="budget for the numpy cpu_baseline leg (0 disables it)")
    ap.add_argument("--parity-frames", type=int, default=32)
    ap.add_argument("--cpu-all-cores", action="store_true",
                    help="also time the oracle with one forked worker per visible host core (BASELINE.md §3; "
                         "pure CPU, runs before any GPU runtime is loaded; do not use under rocprofv3)")
    return ap.parse_args()


def cpu_baseline(window: str, seconds: float):
    """The oracle (oracle/cpu_ref.py, the numpy restatement of streamer.py:119-121) on a
    bounded sample of the same workload, single host core.  A reported baseline only."""
    import numpy as np
    from oracle import cpu_ref
    from sdr_iq_visualizer_amd import synth

    chunk, n_chunks = 128, 16                       # 2^11 frames = 8.4 Msamples per pass
    frames = [synth.synth_iq(1234, c * chunk, chunk, NFFT) for c in range(n_chunks)]
    w = np.hanning(NFFT).astype(np.float32) if window == "hann" else None
    cpu_ref.spectrum_db(frames[0], window=w)        # warm-up
    done, t0 = 0, time.perf_counter()
    while True:
        for f in frames:
            cpu_ref.spectrum_db(f, window=w)
        done += chunk * n_chunks
        dt = time.perf_counter() - t0
        if dt >= seconds:
            break
    batched = done * NFFT / dt / 1e6
    # the reference's literal per-frame expression (rectangular, as it ships)
    t0, k = time.perf_counter(), 0
    while time.perf_counter() - t0 < min(2.0, seconds / 4):
        for f in frames[0]:
            20 * np.log10(np.abs(np.fft.fftshift(np.fft.fft(f))) + 1e-12)
        k += chunk
    per_frame = k * NFFT / (time.perf_counter() - t0) / 1e6
    return {
        "value": round(batched, 2),
        "unit": "Msamples/s",
        "cores": 1,
        "kind": "port",
        "sample": f"{done} frames x {NFFT} ({window}), oracle/cpu_ref.spectrum_db in chunks of {chunk} "
                  f"for {dt:.1f} s, numpy {np.__version__}, {os.cpu_count()} host cores visible",
        "per_frame_reference_expression_rect": round(per_frame, 2),
    }


def _cpu_worker(job):
    """One forked worker of the all-cores leg: the batched oracle on its own frames for `seconds`."""
    import numpy as np
    from oracle import cpu_ref
    from sdr_iq_visualizer_amd import synth
    seed, window, seconds = job
    frames = [synth.synth_iq(seed, c * 128, 128, NFFT) for c in range(4)]
    w = np.hanning(NFFT).astype(np.float32) if window == "hann" else None
    cpu_ref.spectrum_db(frames[0], window=w)
    done, t0 = 0, time.perf_counter()
    while time.perf_counter() - t0 < seconds:
        for f in frames:
            cpu_ref.spectrum_db(f, window=w)
        done += 128 * len(frames)
    return done, time.perf_counter() - t0


def cpu_baseline_all_cores(window: str, seconds: float):
    import multiprocessing as mp
    try:
        cores = len(os.sched_getaffinity(0))
    except AttributeError:
        cores = os.cpu_count() or 1
    with mp.get_context("fork").Pool(cores) as pool:
        res = pool.map(_cpu_worker, [(1234 + i, window, seconds) for i in range(cores)])
    total = sum(r[0] for r in res) * NFFT / max(r[1] for r in res) / 1e6
    quota = None
    try:
        q, per = open("/sys/fs/cgroup/cpu.max").read().split()
        quota = None if q == "max" else round(int(q) / int(per), 1)
    except Exception:
        pass
    return {"value": round(total, 1), "unit": "Msamples/s", "workers": cores, "cgroup_cpu_quota": quota}


def main():
    args = parse_args()
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            sys.exit("bench.py --gpus N>1 must be launched with python -m torch.distributed.run "
                     "--nproc-per-node N (one rank per GPU)")
        args.gpus = world

    # cpu baseline first: plain numpy, before any GPU runtime is up (rank 0, N=1 only)
    cpu = None
    if rank == 0 and world == 1 and args.cpu_seconds > 0:
        cpu = cpu_baseline(args.window, args.cpu_seconds)
        if args.cpu_all_cores:
            cpu["all_cores"] = cpu_baseline_all_cores(args.window, min(args.cpu_seconds, 5.0))

    import torch  # before libsdrk: one shared HIP runtime in the process (see _ffi.py)
    import numpy as np
    from oracle import cpu_ref
    from sdr_iq_visualizer_amd import _ffi, synth
    from sdr_iq_visualizer_amd.spectrum import SpectrumPlan

    dist = None
    n_dev = torch.cuda.device_count()
    dev = local_rank % max(n_dev, 1)
    # One rank per GPU over RCCL ("nccl").  Rehearsal on a box with fewer GPUs than ranks
    # (ranks then share a device, which RCCL refuses): gloo carries the barrier/max-reduce.
    backend = "nccl" if n_dev >= world else "gloo"
    if "RANK" in os.environ:
        import torch.distributed as dist
        torch.cuda.set_device(dev)
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", dev))
        else:
            dist.init_process_group("gloo")
    red_dev = "cuda" if backend == "nccl" else "cpu"
    lib = _ffi.lib()
    _ffi.require_device(dev)

    frames = args.frames
    first_frame = rank * frames                     # config 4: GPU g owns [g*F, (g+1)*F)
    d_in, d_out = ctypes.c_void_p(), ctypes.c_void_p()
    _ffi.check(lib.sdrk_dev_alloc(dev, frames * NFFT * 8, ctypes.byref(d_in)))
    _ffi.check(lib.sdrk_dev_alloc(dev, frames * NFFT * 4, ctypes.byref(d_out)))
    _ffi.check(lib.sdrk_synth_fill(dev, 1234, first_frame, frames, NFFT, d_in, None))

    plan = SpectrumPlan(NFFT, window=None if args.window == "rect" else args.window, device=dev)

    def barrier():
        plan.sync()
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()

    for _ in range(args.warmup):
        plan.exec_device(d_in.value, frames, d_out.value)
    barrier()
    t0 = time.perf_counter()
    # K steps, bracketed by HIP events on the stream the kernel runs on
    kernel_ms = plan.exec_device_timed(d_in.value, frames, d_out.value, launches=args.steps)
    barrier()
    elapsed = time.perf_counter() - t0
    if dist is not None:
        t = torch.tensor([elapsed, kernel_ms], device=red_dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed, kernel_ms = float(t[0]), float(t[1])

    # parity spot check on this rank's output (oracle = checker only)
    parity = 0.0
    if args.parity_frames > 0:
        rng = np.random.default_rng(rank)
        picks = np.unique(np.concatenate([[0, frames - 1], rng.integers(0, frames, args.parity_frames)]))
        w = np.hanning(NFFT) if args.window == "hann" else None
        row = np.empty(NFFT, dtype=np.float32)
        for f in picks:
            _ffi.check(lib.sdrk_memcpy_d2h(dev, row.ctypes.data_as(ctypes.c_void_p),
                                           ctypes.c_void_p(d_out.value + int(f) * NFFT * 4), row.nbytes))
            ref = cpu_ref.spectrum_db(synth.synth_iq(1234, first_frame + int(f), 1, NFFT)[0], window=w)
            mg, mr = 10.0 ** (row.astype(np.float64) / 20), 10.0 ** (ref.astype(np.float64) / 20)
            parity = max(parity, float(np.abs(mg - mr).max() / mr.max()))
    if dist is not None:
        t = torch.tensor([parity], device=red_dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        parity = float(t[0])

    info = _ffi.device_info(dev)
    plan.close()
    lib.sdrk_dev_free(dev, d_in)
    lib.sdrk_dev_free(dev, d_out)

    if rank == 0:
        samples_per_step = frames * NFFT * world
        ms_per_step = elapsed * 1e3 / args.steps
        value = samples_per_step / (elapsed / args.steps) / 1e6
        launch_ms = kernel_ms / args.steps
        achieved = ALGO_BYTES_PER_SAMPLE * frames * NFFT / (launch_ms * 1e-3) / 1e9   # per GPU
        traffic = None
        tpath = os.path.join(REPO, "profiles", "hbm_traffic.json")
        if os.path.exists(tpath):                    # PMC-derived bytes per launch, see profiles/README.md
            try:
                rec = json.load(open(tpath))
                if rec.get("frames") == frames and rec.get("window") == args.window:
                    traffic = rec.get("bytes_per_launch")
            except Exception:
                traffic = None
        line = {
            "metric": METRIC,
            "value": round(value, 1),
            "unit": "Msamples/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": round(ms_per_step, 4),
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f32",
            "data": "synthetic",
            "config": {
                "workload": f"batched N={NFFT} {args.window}-windowed FFT + fftshift + 20*log10(|X|+1e-12), "
                            f"{frames} device-resident complex64 frames per GPU (BASELINE.json configs[1]"
                            + ("; configs[3] frame-range sharding" if world > 1 else "") + ")",
                "nfft": NFFT, "frames_per_gpu": frames, "window": args.window,
                "sharding": f"frame-range x{world}, no collectives", "device": info,
                "rendezvous_backend": (backend if dist is not None else None),
            },
            "hbm_peak_frac": round(value * 1e6 * ALGO_BYTES_PER_SAMPLE / 1e9 / (HBM_PEAK_GBPS * world), 4),
            "roofline": {
                "bound": "hbm",
                "achieved": round(achieved, 1),
                "peak": HBM_PEAK_GBPS,
                "unit": "GB/s",
                "frac": round(achieved / HBM_PEAK_GBPS, 4),
                "traffic": traffic,
                "kernel": "sdrk::fft4096_kernel",
                "kernel_ms_per_launch": round(launch_ms, 4),
                "algorithmic_bytes_per_launch": ALGO_BYTES_PER_SAMPLE * frames * NFFT,
            },
            "parity_max_rel_err": parity,
        }
        if cpu is not None:
            line["cpu_baseline"] = cpu
        print(json.dumps(line), flush=True)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
