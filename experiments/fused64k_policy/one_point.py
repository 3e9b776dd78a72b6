#!/usr/bin/env python3
"""One point of the sweep in its own process (for rocprofv3 passes): the policy comes from the SDRK_FU_* environment the caller
set; `tiled` as third argument runs the two tiled launches instead.   one_point.py [frames] [hop] [fused|tiled] [launches]"""
import ctypes
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from sdr_iq_visualizer_amd import _ffi  # noqa: E402
from sdr_iq_visualizer_amd.spectrum import SpectrumPlan  # noqa: E402

n = 65536
nf = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
hop = int(sys.argv[2]) if len(sys.argv) > 2 else n
kind = sys.argv[3] if len(sys.argv) > 3 else "fused"
k = int(sys.argv[4]) if len(sys.argv) > 4 else 4
lib = _ffi.lib()
samples = (nf - 1) * hop + n
d_in, d_b = ctypes.c_void_p(), ctypes.c_void_p()
_ffi.check(lib.sdrk_dev_alloc(0, samples * 8, ctypes.byref(d_in)))
_ffi.check(lib.sdrk_dev_alloc(0, nf * n * 4, ctypes.byref(d_b)))
_ffi.check(lib.sdrk_synth_fill(0, 3, 0, (samples + 4095) // 4096, 4096, d_in, None))
with SpectrumPlan(n, window="hann", fused64k=(kind == "fused")) as p:   # False = the two tiled launches
    t0 = time.perf_counter()
    while (time.perf_counter() - t0) < 0.04:
        p.exec_device_timed(d_in.value, nf, d_b.value, 1, frame_stride=hop)
    ms = p.exec_device_timed_each(d_in.value, nf, d_b.value, k, frame_stride=hop)
    p.sync()
print(kind, {v: os.environ.get(v) for v in ("SDRK_FU_WG_PER_CU", "SDRK_FU_IN_AUX", "SDRK_FU_OUT_AUX", "SDRK_FU_NOWAIT")},
      "ms each:", [round(m, 3) for m in ms])
