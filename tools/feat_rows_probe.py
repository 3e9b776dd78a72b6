#!/usr/bin/env python3
"""Developer helper (GPU box): the stand-alone per-row reductions on device-resident rows of several lengths (noise rows,
the worst case for the peak list).  python3 tools/feat_rows_probe.py"""
import ctypes, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from sdr_iq_visualizer_amd import _ffi, features
from sdr_iq_visualizer_amd.spectrum import SpectrumPlan
lib = _ffi.lib()
for n, nf in ((1024, 1 << 16), (4096, 1 << 15), (8192, 1 << 14), (16384, 1 << 13), (32768, 1 << 12), (65536, 1 << 11), (1 << 20, 128)):
    mp = 64
    b = {}
    gen = (nf * n + 4095) // 4096                      # generator frames of 4096 samples covering the input exactly
    for name, sz in (("iq", gen * 4096 * 8), ("rows", nf * n * 4)):
        b[name] = ctypes.c_void_p(); _ffi.check(lib.sdrk_dev_alloc(0, sz, ctypes.byref(b[name])))
    _ffi.check(lib.sdrk_synth_fill(0, 4321, 0, gen, 4096, b["iq"], None))
    with SpectrumPlan(n, window="hann") as plan:
        plan.exec_device(b["iq"].value, nf, b["rows"].value); plan.sync()
    rank, gamma = features.percentile_rank(n, 20.0), float(features.percentile_gamma(n, 20.0))
    host = {"stats": np.empty((nf, 16)), "thr": np.empty(nf), "idx": np.empty((nf, mp), dtype=np.int32), "cnt": np.empty(nf, dtype=np.int32)}
    hp = {k: v.ctypes.data_as(ctypes.c_void_p) for k, v in host.items()}
    def run(peaks=True):
        _ffi.check(lib.sdrk_row_features(0, b["rows"], 1, nf, n, rank, ctypes.c_float(gamma), max(3, n // 300), mp, hp["stats"], hp["thr"],
                                         hp["idx"] if peaks else None, hp["cnt"] if peaks else None))
    run()
    t0 = time.perf_counter(); run(False); dt_np = time.perf_counter() - t0
    t0 = time.perf_counter(); run(); dt = time.perf_counter() - t0
    print(f"  (without the peak list {dt_np*1e3:8.3f} ms)", end="")
    print(f"n={n:8d} rows={nf:6d}: {dt*1e3:8.3f} ms  {dt/nf*1e6:8.2f} us/row  {nf*n*4/dt/1e9:7.1f} GB/s of rows  peaks/row {host['cnt'].mean():.0f}", flush=True)
    for v in b.values(): lib.sdrk_dev_free(0, v)
