"""Developer helper (GPU box): latency of the live app's call shape, one 4096-sample frame host -> host."""
import sys, time
import numpy as np
sys.path.insert(0, __file__.rsplit("/", 2)[0])
import sdr_iq_visualizer_amd as pkg
from sdr_iq_visualizer_amd import synth
x = synth.synth_iq(1, 0, 1, 4096)[0]
for _ in range(200): pkg.spectrum_db(x)


def ref3():   # the three lines of app/sdr/streamer.py:119-121 as they stand
    fft_data = np.fft.fftshift(np.fft.fft(x))
    freqs = np.fft.fftshift(np.fft.fftfreq(len(x), 1 / 1e6)) + 2.4e9
    power_db = 20 * np.log10(np.abs(fft_data) + 1e-12)
    return freqs, power_db


for name, fn in (("spectrum_db(x)", lambda: pkg.spectrum_db(x)), ("process_frame", lambda: pkg.process_frame(x, 1e6, 2.4e9)),
                 ("numpy expression", lambda: 20 * np.log10(np.abs(np.fft.fftshift(np.fft.fft(x))) + 1e-12)),
                 ("numpy 3 lines", ref3)):
    ts = []
    for _ in range(3000):
        t0 = time.perf_counter(); fn(); ts.append(time.perf_counter() - t0)
    ts.sort()
    print(f"{name:18s} median {ts[len(ts)//2]*1e6:6.1f} us   p10 {ts[len(ts)//10]*1e6:6.1f}   p90 {ts[9*len(ts)//10]*1e6:6.1f}")
import ctypes
from sdr_iq_visualizer_amd import _ffi
from sdr_iq_visualizer_amd.spectrum import SpectrumPlan
lib = _ffi.lib()
with SpectrumPlan(4096) as p:
    out = np.empty(4096, np.float32)
    a, b, h = x.ctypes.data_as(ctypes.c_void_p), out.ctypes.data_as(ctypes.c_void_p), p.handle
    f = lib.sdrk_exec_host
    for _ in range(200): f(h, a, 1, 4096, b)
    ts = []
    for _ in range(3000):
        t0 = time.perf_counter(); f(h, a, ctypes.c_size_t(1), ctypes.c_size_t(4096), b); ts.append(time.perf_counter() - t0)
    ts.sort()
    print(f"{'C call via ctypes':18s} median {ts[len(ts)//2]*1e6:6.1f} us   p10 {ts[len(ts)//10]*1e6:6.1f}   p90 {ts[9*len(ts)//10]*1e6:6.1f}")
