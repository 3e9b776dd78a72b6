"""Developer helper (GPU box): N = 65536, the fused single-launch kernel against the two tiled launches.
    python tools/fused_probe.py [frames] [hop] [window]
Prints the median of 7 timed executions of each (after 100 ms of warm-up) and the largest difference between their outputs."""
import ctypes
import sys
import time

import numpy as np

sys.path.insert(0, __file__.rsplit("/", 2)[0])
from sdr_iq_visualizer_amd import _ffi  # noqa: E402
from sdr_iq_visualizer_amd.spectrum import SpectrumPlan  # noqa: E402

n = 65536
nf = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
hop = int(sys.argv[2]) if len(sys.argv) > 2 else n
window = sys.argv[3] if len(sys.argv) > 3 else "hann"
lib = _ffi.lib()
samples = (nf - 1) * hop + n
d_in, d_a, d_b = ctypes.c_void_p(), ctypes.c_void_p(), ctypes.c_void_p()
_ffi.check(lib.sdrk_dev_alloc(0, samples * 8, ctypes.byref(d_in)))
_ffi.check(lib.sdrk_dev_alloc(0, nf * n * 4, ctypes.byref(d_a)))
_ffi.check(lib.sdrk_dev_alloc(0, nf * n * 4, ctypes.byref(d_b)))
_ffi.check(lib.sdrk_synth_fill(0, 3, 0, (samples + 4095) // 4096, 4096, d_in, None))
res = {}
for name, kw, d_out in (("tiled", {"fused64k": False}, d_a), ("fused", {"fused64k": True}, d_b)):
    with SpectrumPlan(n, window=None if window == "rect" else window, **kw) as p:
        t0 = time.perf_counter()                     # warm up BY TIME (an idle device needs tens of ms to reach its clock)
        while time.perf_counter() - t0 < 0.1:
            p.exec_device_timed(d_in.value, nf, d_out.value, 1, frame_stride=hop)
        ms = sorted(p.exec_device_timed_each(d_in.value, nf, d_out.value, 7, frame_stride=hop))
        p.sync()
        alg = (8 * samples + 4 * nf * n) / ms[3] / 1e6
        print(f"{name}: median {ms[3]:.3f} ms (min {ms[0]:.3f}) for {nf} frames hop {hop}: {alg:.0f} GB/s algorithmic, "
              f"{alg / 8000:.3f} of peak")
    res[name] = ms[3]
rows = min(nf, 64)
a = np.empty(rows * n, np.float32)
b = np.empty(rows * n, np.float32)
for off in (0, (nf - rows) * n * 4):
    _ffi.check(lib.sdrk_memcpy_d2h(0, a.ctypes.data_as(ctypes.c_void_p), ctypes.c_void_p(d_a.value + off), a.nbytes))
    _ffi.check(lib.sdrk_memcpy_d2h(0, b.ctypes.data_as(ctypes.c_void_p), ctypes.c_void_p(d_b.value + off), b.nbytes))
    print("rows at byte offset", off, "identical:", bool(np.array_equal(a, b)), "max abs diff", float(np.max(np.abs(a - b))))
print(f"fused / tiled time: {res['fused'] / res['tiled']:.3f}")
