#!/usr/bin/env python3
"""Where does the persistent N = 65536 launch start to pay?  Forced fused against forced tiled for calls of 1 ... 4096 frames,
device resident, warm, median of 15 launches (events between consecutive launches).  Sets FUSED_AUTO_MIN_FRAMES (sdrk_api.hip).
    crossover.py [hop]"""
import ctypes
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from sdr_iq_visualizer_amd import _ffi  # noqa: E402
from sdr_iq_visualizer_amd.spectrum import SpectrumPlan  # noqa: E402

n = 65536
hop = int(sys.argv[1]) if len(sys.argv) > 1 else 32768
lib = _ffi.lib()
nmax = 4096
samples = (nmax - 1) * hop + n
d_in, d_b = ctypes.c_void_p(), ctypes.c_void_p()
_ffi.check(lib.sdrk_dev_alloc(0, samples * 8, ctypes.byref(d_in)))
_ffi.check(lib.sdrk_dev_alloc(0, nmax * n * 4, ctypes.byref(d_b)))
_ffi.check(lib.sdrk_synth_fill(0, 3, 0, (samples + 4095) // 4096, 4096, d_in, None))
print(f"# hop {hop}: frames per call, tiled us, fused us, fused / tiled")
with SpectrumPlan(n, window="hann", fused64k=True) as pf, SpectrumPlan(n, window="hann", fused64k=False) as pt:
    for nf in (1, 4, 8, 16, 24, 32, 48, 64, 96, 128, 192, 256, 384, 512, 768, 1024, 2048, 4096):
        res = []
        for p in (pt, pf):
            t0 = time.perf_counter()
            while (time.perf_counter() - t0) < 0.05:
                p.exec_device_timed(d_in.value, nf, d_b.value, 4, frame_stride=hop)
            ms = sorted(p.exec_device_timed_each(d_in.value, nf, d_b.value, 15, frame_stride=hop))
            res.append(ms[7] * 1e3)
        print(f"{nf:6d}  {res[0]:9.1f}  {res[1]:9.1f}  {res[1] / res[0]:6.3f}", flush=True)
for d in (d_in, d_b):
    lib.sdrk_dev_free(0, d)
