import sys, os, ctypes
sys.path.insert(0, '.')
import numpy as np
from sdr_iq_visualizer_amd import _ffi
from sdr_iq_visualizer_amd.spectrum import SpectrumPlan
lib = _ffi.lib()
n, nf = 65536, 600
d_in, d_a, d_b = ctypes.c_void_p(), ctypes.c_void_p(), ctypes.c_void_p()
_ffi.check(lib.sdrk_dev_alloc(0, nf * n * 8, ctypes.byref(d_in)))
_ffi.check(lib.sdrk_dev_alloc(0, nf * n * 4, ctypes.byref(d_a)))
_ffi.check(lib.sdrk_dev_alloc(0, nf * n * 4, ctypes.byref(d_b)))
_ffi.check(lib.sdrk_synth_fill(0, 77, 0, nf * 16, 4096, d_in, None))
os.environ["SDRK_FUSED64K"] = "1"; fused = SpectrumPlan(n, window="hann")
os.environ["SDRK_FUSED64K"] = "0"; tiled = SpectrumPlan(n, window="hann")
tiled.exec_device(d_in.value, nf, d_b.value); tiled.sync()
B = np.empty((nf, n), dtype=np.uint32)
_ffi.check(lib.sdrk_memcpy_d2h(0, B.ctypes.data_as(ctypes.c_void_p), d_b, B.nbytes))
for rep in range(3):
    fused.exec_device(d_in.value, nf, d_a.value); fused.sync()
    A = np.empty((nf, n), dtype=np.uint32)
    _ffi.check(lib.sdrk_memcpy_d2h(0, A.ctypes.data_as(ctypes.c_void_p), d_a, A.nbytes))
    bad = np.argwhere(A != B)
    print("rep", rep, "mismatching elements:", len(bad), "frames:", len(np.unique(bad[:, 0])) if len(bad) else 0)
    if len(bad):
        f = bad[0, 0]
        idx = bad[bad[:, 0] == f][:, 1]
        k3 = idx % 256; k1 = idx // 256
        print(" frame", f, "n bad", len(idx), "k3 values", np.unique(k3)[:40], "k1 values", np.unique(k1 ^ 128)[:40], len(np.unique(k1)))
