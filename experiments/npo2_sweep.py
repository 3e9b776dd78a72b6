import sys, time, ctypes
sys.path.insert(0, ".")
from sdr_iq_visualizer_amd import _ffi
from sdr_iq_visualizer_amd.spectrum import SpectrumPlan
lib=_ffi.lib()
for n in (20000, 50000, 65537, 100000, 300000, 1000000):
    nf = max(1, (1<<27)//n)
    d_in, d_out = ctypes.c_void_p(), ctypes.c_void_p()
    _ffi.check(lib.sdrk_dev_alloc(0, nf*n*8, ctypes.byref(d_in))); _ffi.check(lib.sdrk_dev_alloc(0, nf*n*4, ctypes.byref(d_out)))
    _ffi.check(lib.sdrk_synth_fill(0, 3, 0, (nf*n+4095)//4096, 4096, d_in, None))
    with SpectrumPlan(n) as p:
        t0=time.perf_counter()
        while time.perf_counter()-t0 < 0.1: p.exec_device_timed(d_in.value, nf, d_out.value, 1)
        ms=sorted(p.exec_device_timed_each(d_in.value, nf, d_out.value, 7))[3]
    print(n, nf, round(ms,3), "ms", round(12*nf*n/ms/1e6/8000,4), "of peak")
    lib.sdrk_dev_free(0,d_in); lib.sdrk_dev_free(0,d_out)
