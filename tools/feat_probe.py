#!/usr/bin/env python3
"""Developer helper: time the per-row reductions (fused N=4096 epilogue and stand-alone kernel), with and without the peak scan."""
import ctypes, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from sdr_iq_visualizer_amd import _ffi, features
from sdr_iq_visualizer_amd.spectrum import SpectrumPlan
lib = _ffi.lib()
n, nf, mp = 4096, 1 << 16, 64
rank, gamma = features.percentile_rank(n, 20.0), float(features.percentile_gamma(n, 20.0))
b = {}
for name, sz in (("iq", nf * n * 8), ("rows", nf * n * 4), ("stats", nf * 128), ("thr", nf * 8), ("idx", nf * mp * 4), ("cnt", nf * 4)):
    b[name] = ctypes.c_void_p(); _ffi.check(lib.sdrk_dev_alloc(0, sz, ctypes.byref(b[name])))
_ffi.check(lib.sdrk_synth_fill(0, 4321, 0, nf, n, b["iq"], None))
with SpectrumPlan(n, window="hann") as plan:
    def run(peaks, rows):
        _ffi.check(lib.sdrk_frame_features_device(plan.handle, b["iq"], nf, n, b["rows"] if rows else None, rank, ctypes.c_float(gamma),
                                                  13, mp, b["stats"], b["thr"], b["idx"] if peaks else None, b["cnt"] if peaks else None, None))
        plan.sync()
    def timed(fn, warm_ms=100.0, reps=7):        # warm by time: an idle device needs tens of ms of load to reach its sustained clock
        t0 = time.perf_counter()
        while (time.perf_counter() - t0) * 1e3 < warm_ms:
            fn()
        ts = []
        for _ in range(reps):
            t0 = time.perf_counter(); fn(); ts.append(time.perf_counter() - t0)
        return sorted(ts)[len(ts) // 2]
    for peaks in (True, False):
        run(peaks, False)
        t0 = time.perf_counter(); run(peaks, False); dt = time.perf_counter() - t0
        print(f"fused peaks={peaks}: cold (second launch after idle) {dt*1e3:.3f} ms, {dt/nf*1e9:.1f} ns/row", flush=True)
        dt = timed(lambda: run(peaks, False))
        print(f"fused peaks={peaks}: warm {dt*1e3:.3f} ms, {dt/nf*1e9:.1f} ns/row = {nf/dt/1e6:.1f} M rows/s", flush=True)
    plan.exec_device(b["iq"].value, nf, b["rows"].value); plan.sync()
    t0 = time.perf_counter(); plan.exec_device(b["iq"].value, nf, b["rows"].value); plan.sync(); print(f"transform only: {(time.perf_counter()-t0)*1e3:.3f} ms")
    # the stand-alone single-read reduction kernel over rows already in HBM (results to the host, as sdrk_row_features does)
    host = {"stats": np.empty((nf, 16)), "thr": np.empty(nf), "idx": np.empty((nf, mp), dtype=np.int32), "cnt": np.empty(nf, dtype=np.int32)}
    hp = {k: v.ctypes.data_as(ctypes.c_void_p) for k, v in host.items()}
    def alone():
        _ffi.check(lib.sdrk_row_features(0, b["rows"], 1, nf, n, rank, ctypes.c_float(gamma), 13, mp, hp["stats"], hp["thr"], hp["idx"], hp["cnt"]))
    alone()
    t0 = time.perf_counter(); alone(); dt = time.perf_counter() - t0
    print(f"stand-alone kernel + D2H of results: {dt*1e3:.3f} ms, {dt/nf*1e9:.1f} ns/row", flush=True)
cnt = np.empty(nf, dtype=np.int32)
_ffi.check(lib.sdrk_memcpy_d2h(0, cnt.ctypes.data_as(ctypes.c_void_p), b["cnt"], cnt.nbytes))
print("peaks per row: mean", cnt.mean(), "max", cnt.max())
