#!/usr/bin/env python3
"""Developer helper: the numpy boundary at a few (nfft, batch) points with SDRK_HOST_TRACE=1 (stderr shows the
phases of each sdrk_exec_host call).  usage: host_sweep.py [nfft:batch ...]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("SDRK_HOST_TRACE", "1")
import numpy as np
import sdr_iq_visualizer_amd as pkg
from sdr_iq_visualizer_amd import synth
for arg in (sys.argv[1:] or ["4096:4096", "4096:16384"]):
    n, b = (int(v) for v in arg.split(":"))
    x = synth.synth_iq(1, 0, b * n // 4096, 4096).reshape(b, n)
    for rep in range(3):
        t0 = time.perf_counter(); y = pkg.spectrum_db(x); dt = time.perf_counter() - t0
        print(f"N={n} B={b} call {rep}: {dt*1e3:.3f} ms, {b*n*8/dt/1e9:.1f} GB/s in", flush=True)
