#!/usr/bin/env python3
"""Developer helper (GPU box): the fused N = 4096 epilogue against the stand-alone reduction kernel on the same frames, column
by column of the packed results (bench.py only reports whether ALL of them agree)."""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from sdr_iq_visualizer_amd import _ffi, features
from sdr_iq_visualizer_amd.spectrum import SpectrumPlan
lib = _ffi.lib()
n, nf, mp = 4096, 1 << 15, 64
rank, gamma = features.percentile_rank(n, 20.0), float(features.percentile_gamma(n, 20.0))
b = {}
for name, sz in (("iq", nf * n * 8), ("rows", nf * n * 4), ("rows2", nf * n * 4), ("stats", nf * 128), ("thr", nf * 8), ("idx", nf * mp * 4), ("cnt", nf * 4)):
    b[name] = ctypes.c_void_p(); _ffi.check(lib.sdrk_dev_alloc(0, sz, ctypes.byref(b[name])))
_ffi.check(lib.sdrk_synth_fill(0, 4321, 0, nf, n, b["iq"], None))
def d2h(ptr, shape, dt):
    a = np.empty(shape, dtype=dt); _ffi.check(lib.sdrk_memcpy_d2h(0, a.ctypes.data_as(ctypes.c_void_p), ptr, a.nbytes)); return a
with SpectrumPlan(n, window="hann") as plan:
    _ffi.check(lib.sdrk_frame_features_device(plan.handle, b["iq"], nf, n, b["rows"], rank, ctypes.c_float(gamma), 13, mp, b["stats"], b["thr"], b["idx"], b["cnt"], None))
    plan.sync()
    fs, ft, fi, fc, frows = d2h(b["stats"], (nf, 16), np.float64), d2h(b["thr"], nf, np.float64), d2h(b["idx"], (nf, mp), np.int32), d2h(b["cnt"], nf, np.int32), d2h(b["rows"], (nf, n), np.float32)
    plan.exec_device(b["iq"].value, nf, b["rows2"].value); plan.sync()
    prow = d2h(b["rows2"], (nf, n), np.float32)
    print("rows of the fused kernel == rows of fft4096_kernel:", np.array_equal(frows, prow), "differing rows:", int((frows != prow).any(axis=1).sum()))
    for src, nm in ((b["rows"], "fused kernel's rows"), (b["rows2"], "fft4096_kernel's rows")):
        hs, ht, hi, hc = np.empty((nf, 16)), np.empty(nf), np.empty((nf, mp), dtype=np.int32), np.empty(nf, dtype=np.int32)
        _ffi.check(lib.sdrk_row_features(0, src, 1, nf, n, rank, ctypes.c_float(gamma), 13, mp, hs.ctypes.data_as(ctypes.c_void_p), ht.ctypes.data_as(ctypes.c_void_p), hi.ctypes.data_as(ctypes.c_void_p), hc.ctypes.data_as(ctypes.c_void_p)))
        print(f"stand-alone kernel over the {nm}:")
        for c in range(16):
            bad = np.flatnonzero(~((fs[:, c] == hs[:, c]) | (np.isnan(fs[:, c]) & np.isnan(hs[:, c]))))
            if bad.size: print(f"   stats[{c}]: {bad.size} rows differ, e.g. row {bad[0]}: fused {fs[bad[0], c]!r} alone {hs[bad[0], c]!r}")
        print("   thr differ:", int((ft != ht).sum()), " cnt differ:", int((fc != hc).sum()))
