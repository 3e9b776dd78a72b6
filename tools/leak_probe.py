"""Developer helper (GPU box): device memory before/after repeated create-use-destroy cycles, per object kind."""
import ctypes, sys
import numpy as np
sys.path.insert(0, __file__.rsplit("/", 2)[0])
import sdr_iq_visualizer_amd as pkg
from sdr_iq_visualizer_amd import _ffi, features
from sdr_iq_visualizer_amd.spectrum import SpectrumPlan

def free_bytes():
    free, total = ctypes.c_size_t(), ctypes.c_size_t()
    _ffi.check(_ffi.lib().sdrk_dev_mem_info(0, ctypes.byref(free), ctypes.byref(total)))
    return free.value

rng = np.random.default_rng(1)
def r(*s): return (rng.standard_normal(s) + 1j * rng.standard_normal(s)).astype(np.complex64)
x = {n: r(2, n) for n in (4096, 1000, 8192, 65536)}
big = r(1, 1 << 20)
def plan(n, **kw):
    def f():
        with SpectrumPlan(n, **kw) as p:
            p.spectrum_db(x.get(n, big))
    return f
def wf():
    w = pkg.WaterfallBuffer(4096, maxlen=100); w.append(x[4096][0]); w.as_array(); w.close()
def wf_async():
    with pkg.WaterfallBuffer(4096, maxlen=8, window="hann") as w:
        d = ctypes.c_void_p()
        _ffi.check(_ffi.lib().sdrk_dev_alloc(0, x[4096].nbytes, ctypes.byref(d)))
        _ffi.check(_ffi.lib().sdrk_memcpy_h2d(0, d, x[4096].ctypes.data_as(ctypes.c_void_p), x[4096].nbytes))
        w.append_iq_device(d.value, 2, wait=False); w.gather_begin(None, decimate=16); w.gather_end(); w.sync()
        _ffi.check(_ffi.lib().sdrk_dev_free(0, d))
batch = r(1500, 4096)                                    # 48 MiB: the pipelined form of sdrk_frame_features_host
def plan_batch():
    with SpectrumPlan(4096, window="hann") as p:
        p.spectrum_db(batch)
def pinned():
    a = pkg.pinned_empty((64, 4096), np.complex64); a[...] = 1; pkg.spectrum_db(a); del a
def tuned():
    with SpectrumPlan(4096, window="hann", tune_staging=True) as p:
        p.spectrum_db(batch)
def auto_pin():                                          # a fresh pair of arrays per cycle, page-locked at once (cheap in the reckoning)
    from sdr_iq_visualizer_amd import hostmem
    old, hostmem.COSTS = hostmem.COSTS, hostmem.HostCosts(register_ms_per_GiB=0.001)
    try:
        a, o = batch.copy(), np.empty(batch.shape, np.float32)
        pkg.spectrum_db(a, devices=[0, 0], out=o); pkg.spectrum_db(a, devices=[0, 0], out=o)
        assert pkg.is_pinned(a)
        del a, o
    finally:
        hostmem.COSTS = old
    assert not hostmem._auto_registered
kinds = {"tuned_staging": tuned, "auto_pin": auto_pin, "row_planes": lambda: features.row_features(np.zeros((300, 4096), np.float32), None, as_arrays=True, max_peaks=32),
         "waterfall_async": wf_async, "features_batch": lambda: features.frame_features(batch, 1e6, 2.4e9, max_peaks=16, as_arrays=True),
         "host_pipeline": plan_batch, "pinned_arrays": pinned,
         "plan4096": plan(4096, window="hann"), "plan1000": plan(1000), "plan8192": plan(8192), "plan65536": plan(65536, window="hann"),
         "fused": plan(65536, fused64k=True), "plan2^20": plan(1 << 20), "waterfall": wf,
         "features": lambda: features.frame_features(x[4096], 1e6, 2.4e9)}
for name, f in kinds.items():
    f()
    a = free_bytes()
    for _ in range(16): f()
    b = free_bytes()
    for _ in range(16): f()
    c = free_bytes()
    print(f"{name:10s} after 16: {(a-b)/2**20:8.2f} MiB  after 32: {(a-c)/2**20:8.2f} MiB")
