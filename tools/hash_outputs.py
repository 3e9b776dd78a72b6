#!/usr/bin/env python3
"""Developer helper (GPU box): SHA-1 of the device results of one library build on fixed synthetic input, per frame length and
window — two builds that print the same lines are bit-identical (SDRK_LIB selects the build; tools/variant.sh makes them)."""
import ctypes, hashlib, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from sdr_iq_visualizer_amd import _ffi
from sdr_iq_visualizer_amd.spectrum import SpectrumPlan
lib = _ffi.lib()
for n, nf in ((16, 4096), (64, 4096), (256, 2048), (512, 1024), (1024, 1024), (2048, 512), (4096, 1024), (8192, 256), (16384, 128),
              (32768, 64), (65536, 48), (1 << 17, 16), (1 << 18, 8), (1 << 19, 8), (1 << 20, 4), (1 << 21, 2), (1 << 22, 2), (1000, 64), (5000, 16)):
    for window in (None, "hann"):
        d_in, d_out = ctypes.c_void_p(), ctypes.c_void_p()
        _ffi.check(lib.sdrk_dev_alloc(0, nf * n * 8, ctypes.byref(d_in))); _ffi.check(lib.sdrk_dev_alloc(0, nf * n * 4, ctypes.byref(d_out)))
        with SpectrumPlan(n, window=window, max_batch=nf) as plan:
            _ffi.check(lib.sdrk_synth_fill(0, 77, 0, nf, n, d_in, None))
            plan.exec_device(d_in.value, nf, d_out.value); plan.sync()
            out = np.empty(nf * n, dtype=np.float32)
            _ffi.check(lib.sdrk_memcpy_d2h(0, out.ctypes.data_as(ctypes.c_void_p), d_out, out.nbytes))
        print(n, window, hashlib.sha1(out.tobytes()).hexdigest()[:16], flush=True)
        lib.sdrk_dev_free(0, d_in); lib.sdrk_dev_free(0, d_out)
