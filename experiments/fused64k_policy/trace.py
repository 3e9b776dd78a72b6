#!/usr/bin/env python3
"""Phase timestamps of the fused N = 65536 kernel (library built with -DFU_TRACE; SDRK_LIB selects it): prints, for a col and a row
workgroup of two sets, the mean duration of every phase over frames 8..95 in microseconds (100 MHz clock, 10 ns steps).
    trace.py [frames] [hop]"""
import ctypes
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from sdr_iq_visualizer_amd import _ffi  # noqa: E402
from sdr_iq_visualizer_amd.spectrum import SpectrumPlan  # noqa: E402

n = 65536
nf = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
hop = int(sys.argv[2]) if len(sys.argv) > 2 else n
lib = _ffi.lib()
samples = (nf - 1) * hop + n
d_in, d_b = ctypes.c_void_p(), ctypes.c_void_p()
_ffi.check(lib.sdrk_dev_alloc(0, samples * 8, ctypes.byref(d_in)))
_ffi.check(lib.sdrk_dev_alloc(0, nf * n * 4 + (1 << 20), ctypes.byref(d_b)))
_ffi.check(lib.sdrk_synth_fill(0, 3, 0, (samples + 4095) // 4096, 4096, d_in, None))
with SpectrumPlan(n, window="hann", fused64k=True) as p:
    t0 = time.perf_counter()
    while (time.perf_counter() - t0) < 0.06:
        p.exec_device_timed(d_in.value, nf, d_b.value, 1, frame_stride=hop)
    ms = p.exec_device_timed_each(d_in.value, nf, d_b.value, 6, frame_stride=hop)
    p.sync()
tr = np.empty(2 * 32 * 96 * 8, np.uint32)
_ffi.check(lib.sdrk_memcpy_d2h(0, tr.ctypes.data_as(ctypes.c_void_p), ctypes.c_void_p(d_b.value + nf * n * 4), tr.nbytes))
tr = tr.reshape(2, 32, 96, 8).astype(np.int64)
print(f"{nf} frames hop {hop}: {sorted(ms)[len(ms) // 2]:.3f} ms per launch (traced build)")
col_names = ["top->fft done (prefetch issue, window, transform)", "wait row_done", "issue ring stores", "vmcnt(0): stores + prefetch landed", "flag + register rotate -> next top"]
row_names = ["wait col_done", "issue ring loads", "vmcnt(0): ring loads landed", "flag + transform", "barrier + log epilogue to LDS", "barrier + output stores issued", "barrier -> next top"]
lo, hi = 8, 95
for gi, gname in enumerate(("set 0", "set 7")):
    for role, members, names, slots in (("col", range(0, 16), col_names, 5), ("row", range(16, 32), row_names, 7)):
        t = tr[gi, list(members)]                       # [16][96][8]
        period = float(np.mean((t[:, hi, 0] - t[:, lo, 0]) / (hi - lo))) / 100.0
        print(f"{gname} {role} workgroups (mean over the 16; min..max of the 16 means): period {period:.2f} us per frame")
        for k in range(slots):
            a = t[:, lo:hi, k]
            b = t[:, lo:hi, k + 1] if k + 1 < slots else t[:, lo + 1:hi + 1, 0]
            d = np.mean(b - a, axis=1) / 100.0
            print(f"      {float(np.mean(d)):6.2f} us ({float(d.min()):5.2f} .. {float(d.max()):5.2f})  {names[k]}")
    # the hand-overs of this set, frame by frame: from the LAST flag of one role to the FIRST / LAST wake-up of the other
    c, r = tr[gi, 0:16], tr[gi, 16:32]
    col_flag = c[:, lo:hi, 4].max(axis=0)                       # last col workgroup through its vmcnt(0) for frame s
    col_first = c[:, lo:hi, 4].min(axis=0)
    row_wake = r[:, lo:hi, 1]                                   # rows past their wait for frame s
    row_flag = r[:, lo:hi, 3].max(axis=0)                       # last row workgroup has its loads of frame s
    row_first = r[:, lo:hi, 3].min(axis=0)
    col_wake = c[:, lo + 1:hi + 1, 2]                           # cols past their wait before storing frame s + 1
    print(f"{gname} hand-overs (us): col stores landed first->last workgroup {np.mean(col_flag - col_first) / 100:.2f}; "
          f"last col flag -> first row awake {np.mean(row_wake.min(axis=0) - col_flag) / 100:.2f}, -> last row awake {np.mean(row_wake.max(axis=0) - col_flag) / 100:.2f}")
    print(f"{gname}                   row loads landed first->last workgroup {np.mean(row_flag - row_first) / 100:.2f}; "
          f"last row flag -> first col awake {np.mean(col_wake.min(axis=0) - row_flag) / 100:.2f}, -> last col awake {np.mean(col_wake.max(axis=0) - row_flag) / 100:.2f}")
for d in (d_in, d_b):
    lib.sdrk_dev_free(0, d)
