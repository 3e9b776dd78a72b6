"""Developer helper (GPU box): where a large features.frame_features(as_arrays=True) call spends its time — the C call
(staging / DMA / kernels / results back) against the numpy assembly of the result arrays.  python3 tools/feat_host_probe.py"""
import ctypes, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import sdr_iq_visualizer_amd as pkg
from sdr_iq_visualizer_amd import _ffi, features
from sdr_iq_visualizer_amd.spectrum import SpectrumPlan
lib = _ffi.lib()
n, b, mp = 4096, 1 << 15, 64
x = pkg.pinned_empty((b, n), np.complex64)
x[...] = (np.random.default_rng(5).standard_normal((b, 2 * n), dtype=np.float32) * 200).view(np.complex64)
pageable = np.array(x)
rank, gamma = features.percentile_rank(n, 20.0), float(features.percentile_gamma(n, 20.0))
with SpectrumPlan(n, window="hann") as plan:
    for name, arr in (("pageable", pageable), ("pinned", x)):
        for pinned_out in (False, True):
            mk = (lambda shape, dt: pkg.pinned_empty(shape, dt)) if pinned_out else (lambda shape, dt: np.empty(shape, dt))
            stats, thr, idx, cnt = mk((b, 16), np.float64), mk((b,), np.float64), mk((b, mp), np.int32), mk((b,), np.int32)
            def call():
                _ffi.check(lib.sdrk_frame_features_host(plan.handle, arr.ctypes.data_as(ctypes.c_void_p), b, n, rank, ctypes.c_float(gamma),
                                                        13, mp, stats.ctypes.data_as(ctypes.c_void_p), thr.ctypes.data_as(ctypes.c_void_p),
                                                        idx.ctypes.data_as(ctypes.c_void_p), cnt.ctypes.data_as(ctypes.c_void_p), None))
            call()
            ts = []
            for _ in range(4):
                t0 = time.perf_counter(); call(); ts.append(time.perf_counter() - t0)
            print(f"C call, {name} input, {'pinned' if pinned_out else 'pageable'} results: {sorted(ts)[1]*1e3:.2f} ms = {b*n*8/sorted(ts)[1]/1e9:.1f} GB/s of input", flush=True)
    # round 4: the planes form — the per-row finals formed on the device, the host receives finished arrays
    freqs = pkg.freq_axis(n, 1e6, 0.0)
    for name, arr in (("pageable", pageable), ("pinned", x)):
        for pinned_out in (False, True):
            mk = (lambda shape, dt: pkg.pinned_empty(shape, dt)) if pinned_out else (lambda shape, dt: np.empty(shape, dt))
            planes, idx = mk((_ffi.FEAT_PLANES, b), np.float64), mk((b, mp), np.int32)
            def call():
                _ffi.check(lib.sdrk_frame_features_host_planes(plan.handle, arr.ctypes.data_as(ctypes.c_void_p), b, n, rank,
                                                               ctypes.c_float(gamma), 13, mp, freqs.ctypes.data_as(ctypes.c_void_p),
                                                               planes.ctypes.data_as(ctypes.c_void_p), idx.ctypes.data_as(ctypes.c_void_p), None))
            call()
            ts = []
            for _ in range(4):
                t0 = time.perf_counter(); call(); ts.append(time.perf_counter() - t0)
            print(f"planes C call, {name} input, {'pinned' if pinned_out else 'pageable'} results: {sorted(ts)[1]*1e3:.2f} ms = {b*n*8/sorted(ts)[1]/1e9:.1f} GB/s of input", flush=True)
    for name, arr in (("pageable", pageable), ("pinned", x)):
        features.frame_features(arr[:2048], 1e6, 0.0, window="hann", max_peaks=mp, as_arrays=True)
        ts = []
        for _ in range(4):
            t0 = time.perf_counter(); res = features.frame_features(arr, 1e6, 0.0, window="hann", max_peaks=mp, as_arrays=True); ts.append(time.perf_counter() - t0)
        print(f"features.frame_features(as_arrays=True), {name} input: {sorted(ts)[1]*1e3:.2f} ms = {b*n*8/sorted(ts)[1]/1e9:.1f} GB/s of input", flush=True)
