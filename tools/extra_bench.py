#!/usr/bin/env python3
"""Secondary measurements (not the headline bench): the numpy-in/numpy-out boundary
including PCIe, and the large-frame configs of BASELINE.json (3: N=65536 STFT with 50 %
overlap over 10 s @ 61.44 Msps; 5: N=2^20 frames), device resident, HIP-event timed.

    python tools/extra_bench.py [--quick]
Prints one JSON object per measurement.
"""
from __future__ import annotations

import argparse
import ctypes
import json
import os
import sys
import time

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)

import numpy as np  # noqa: E402

from sdr_iq_visualizer_amd import _ffi, synth  # noqa: E402
from sdr_iq_visualizer_amd.spectrum import SpectrumPlan  # noqa: E402
import sdr_iq_visualizer_amd as pkg  # noqa: E402


def link_probe():
    lib = _ffi.lib()
    a, b, c = ctypes.c_double(), ctypes.c_double(), ctypes.c_double()
    _ffi.check(lib.sdrk_host_link_probe(0, 1 << 30, ctypes.byref(a), ctypes.byref(b), ctypes.byref(c)))
    return {"what": "pinned DMA rates, GB/s (1 GiB up, 0.5 GiB down; duplex quoted on the upstream bytes)",
            "h2d": round(a.value, 1), "d2h": round(b.value, 1), "duplex_up": round(c.value, 1),
            "host_helper_threads": int(lib.sdrk_host_threads())}


def host_boundary(quick):
    out = []
    for b in ((1, 4, 256, 4096) if quick else (1, 4, 8, 16, 64, 256, 1024, 4096, 16384, 32768)):
        x = synth.synth_iq(1, 0, b, 4096)
        pkg.spectrum_db(x)                                  # plan + staging warm-up
        reps = 500 if b <= 16 else 5
        t0 = time.perf_counter()
        for _ in range(reps):
            pkg.spectrum_db(x)
        dt = (time.perf_counter() - t0) / reps
        out.append({"what": "spectrum_db host->host (PCIe inclusive, pageable numpy)", "batch": b, "nfft": 4096,
                    "ms_per_call": round(dt * 1e3, 4), "Msamples_per_s": round(b * 4096 / dt / 1e6, 1),
                    "input_GBps": round(b * 4096 * 8 / dt / 1e9, 2)})
    return out


def device_run(nfft, n_frames, stride, window, reps, label, warm_ms=100.0):
    lib = _ffi.lib()
    in_samples = (n_frames - 1) * stride + nfft
    d_in, d_out = ctypes.c_void_p(), ctypes.c_void_p()
    _ffi.check(lib.sdrk_dev_alloc(0, in_samples * 8, ctypes.byref(d_in)))
    _ffi.check(lib.sdrk_dev_alloc(0, n_frames * nfft * 4, ctypes.byref(d_out)))
    try:
        # fill the stream as consecutive 4096-sample generator frames (length padded up)
        gen_frames = (in_samples + 4095) // 4096
        d_gen = ctypes.c_void_p()
        _ffi.check(lib.sdrk_dev_alloc(0, gen_frames * 4096 * 8, ctypes.byref(d_gen)))
        _ffi.check(lib.sdrk_synth_fill(0, 99, 0, gen_frames, 4096, d_gen, None))
        with SpectrumPlan(nfft, window=window) as plan:
            plan.exec_device(d_gen.value, n_frames, d_out.value, frame_stride=stride)
            plan.sync()
            # warm up BY TIME (round 5): an idle MI355X needs tens of milliseconds of load to reach its sustained shader clock;
            # one warm-up launch + a few isolated ones measured the ramp (config 5: 1.58 ms here against 1.35 ms warm)
            t0 = time.perf_counter()
            while (time.perf_counter() - t0) * 1e3 < warm_ms:
                plan.exec_device_timed(d_gen.value, n_frames, d_out.value, 4, frame_stride=stride)
            ms = sorted(plan.exec_device_timed_each(d_gen.value, n_frames, d_out.value, max(reps, 5), frame_stride=stride))
        _ffi.check(lib.sdrk_dev_free(0, d_gen))
        t = ms[len(ms) // 2] * 1e-3
        algo = 8 * in_samples + 4 * n_frames * nfft
        return {"what": label, "nfft": nfft, "frames": n_frames, "hop": stride, "window": window or "rect",
                "ms": round(t * 1e3, 3), "input_Msamples_per_s": round(in_samples / t / 1e6, 1),
                "frame_Msamples_per_s": round(n_frames * nfft / t / 1e6, 1),
                "algorithmic_GBps": round(algo / t / 1e9, 1), "hbm_peak_frac": round(algo / t / 8e12, 4)}
    finally:
        lib.sdrk_dev_free(0, d_in)
        lib.sdrk_dev_free(0, d_out)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--quick", action="store_true")
    ap.add_argument("--large-only", action="store_true")
    args = ap.parse_args()
    print(json.dumps({"device": pkg.device_info(0)}))
    if not args.large_only:
        print(json.dumps(link_probe()), flush=True)
        for r in host_boundary(args.quick):
            print(json.dumps(r), flush=True)
    L = 614_400_000 if not args.quick else 61_440_000
    n, hop = 65536, 32768
    rows = 1 + (L - n) // hop
    r = device_run(n, rows, hop, "hann", 5, "config 3: STFT N=65536, 50% overlap, 10 s @ 61.44 Msps")
    r["realtime_factor"] = round((L / 61.44e6) / (r["ms"] * 1e-3), 1)
    print(json.dumps(r), flush=True)
    print(json.dumps(device_run(1 << 20, 64 if args.quick else 256, 1 << 20, "hann", 5,
                                "config 5 (one channel): N=2^20 frames back to back")), flush=True)
    print(json.dumps(device_run(65536, 4096, 65536, None, 5, "N=65536 packed frames, rect")), flush=True)
    if not args.large_only:
        print(json.dumps(device_run(4096, 1 << 18, 2048, "hann", 5, "N=4096 STFT 50% overlap")), flush=True)


if __name__ == "__main__":
    main()
