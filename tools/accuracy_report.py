#!/usr/bin/env python3
"""Accuracy of the GPU spectrum vs a float64 reference, next to numpy's own float32 path
(what the reference computes for complex64 input).  rel-L2 and peak-relative max error of the
complex spectrum; max |delta dB| of the log rows on bins within 60 dB of the peak."""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import sdr_iq_visualizer_amd as pkg
from sdr_iq_visualizer_amd import synth

def report(n, frames, window):
    x = synth.synth_iq(2024, 0, frames * max(1, n // 4096), min(n, 4096)).reshape(frames, n) if n >= 4096 else \
        synth.synth_iq(2024, 0, frames, n)
    w = np.hanning(n) if window == "hann" else np.ones(n)
    X64 = np.fft.fft(x.astype(np.complex128) * w, axis=-1)
    Xnp = np.fft.fft((x * w.astype(np.float32)).astype(np.complex64), axis=-1)
    Xg = pkg.fft_c64(x, window=None if window == "rect" else "hann")
    def errs(X):
        d = X.astype(np.complex128) - X64
        return float(np.sqrt((np.abs(d) ** 2).sum() / (np.abs(X64) ** 2).sum())), float((np.abs(d).max(-1) / np.abs(X64).max(-1)).max())
    db64 = 20 * np.log10(np.abs(np.fft.fftshift(X64, axes=-1)) + 1e-12)
    dbg = pkg.spectrum_db(x, window=None if window == "rect" else "hann").astype(np.float64)
    dbn = (20 * np.log10(np.abs(np.fft.fftshift(Xnp, axes=-1)) + 1e-12)).astype(np.float64)
    strong = db64 >= db64.max(-1, keepdims=True) - 60
    return {"nfft": n, "frames": frames, "window": window,
            "gpu_rel_l2": errs(Xg)[0], "numpy_f32_rel_l2": errs(Xnp)[0],
            "gpu_peak_rel_max": errs(Xg)[1], "numpy_f32_peak_rel_max": errs(Xnp)[1],
            "gpu_max_ddb_60": float(np.abs(dbg - db64)[strong].max()), "numpy_f32_max_ddb_60": float(np.abs(dbn - db64)[strong].max())}

for n, frames in ((256, 64), (4096, 64), (65536, 8), (1 << 20, 2)):
    for window in ("rect", "hann"):
        print(json.dumps(report(n, frames, window)), flush=True)
