"""Developer helper (GPU box): the pageable numpy boundary (1 GiB of IQ in, 0.5 GiB of rows out) as a function of the
staging pool's helper threads.  for t in 3 7 11 15; do SDRK_HOST_THREADS=$t python3 tools/host_threads_probe.py; done"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import sdr_iq_visualizer_amd as pkg
b, n = 1 << 15, 4096
x = (np.random.default_rng(1).standard_normal((b, 2 * n), dtype=np.float32)).view(np.complex64)
out = np.empty((b, n), np.float32)
pkg.spectrum_db(x, out=out)
ts = []
for _ in range(5):
    t0 = time.perf_counter(); pkg.spectrum_db(x, out=out); ts.append(time.perf_counter() - t0)
t = sorted(ts)[2]
print(f"helpers {pkg._ffi.lib().sdrk_host_threads()}: {t*1e3:.2f} ms = {b*n*8/t/1e9:.1f} GB/s of input", flush=True)
