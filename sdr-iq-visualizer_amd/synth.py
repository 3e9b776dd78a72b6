"""numpy mirror of the device IQ generator (``sdrk_synth_fill`` in include/sdrk.h).

Inputs only — no spectrum is computed here.  The generator imitates what the
reference reads from the radio at app/sdr/streamer.py:114 (12-bit ADC codes on I
and Q) and is defined on integers so the numpy and HIP versions produce the same
float32 bits:

    F = first_frame + f                                   (64-bit frame number)
    base = fmix32(seed ^ lo32(F)) ^ fmix32(hi32(F) + 0x9E3779B1)
    h = fmix32(base ^ n)                                  (n = sample index in frame)
    I = (h & 0xFFF) - 2048 ;  Q = ((h >> 12) & 0xFFF) - 2048

fmix32 is the MurmurHash3 32-bit finaliser.
"""
from __future__ import annotations

import numpy as np

_M32 = np.uint64(0xFFFFFFFF)


def fmix32(h: np.ndarray) -> np.ndarray:
    """MurmurHash3 finaliser on uint32 values (any shape)."""
    h = np.asarray(h, dtype=np.uint64) & _M32
    h ^= h >> np.uint64(16)
    h = (h * np.uint64(0x85EBCA6B)) & _M32
    h ^= h >> np.uint64(13)
    h = (h * np.uint64(0xC2B2AE35)) & _M32
    h ^= h >> np.uint64(16)
    return h


def synth_iq(seed: int, first_frame: int, n_frames: int, nfft: int) -> np.ndarray:
    """complex64 ``(n_frames, nfft)`` — bit-identical to sdrk_synth_fill."""
    F = np.uint64(first_frame) + np.arange(n_frames, dtype=np.uint64)
    lo, hi = F & _M32, F >> np.uint64(32)
    base = fmix32(np.uint64(seed & 0xFFFFFFFF) ^ lo) ^ fmix32((hi + np.uint64(0x9E3779B1)) & _M32)
    n = np.arange(nfft, dtype=np.uint64)
    h = fmix32(base[:, None] ^ n[None, :])
    i = (h & np.uint64(0xFFF)).astype(np.int64) - 2048
    q = ((h >> np.uint64(12)) & np.uint64(0xFFF)).astype(np.int64) - 2048
    out = np.empty((n_frames, nfft), dtype=np.complex64)
    out.real = i.astype(np.float32)
    out.imag = q.astype(np.float32)
    return out


def tone(nfft: int, k: float, amplitude: float = 1.0, phase: float = 0.0) -> np.ndarray:
    """complex64 exponential at (possibly fractional) bin ``k`` of an nfft frame."""
    n = np.arange(nfft, dtype=np.float64)
    return (amplitude * np.exp(1j * (2.0 * np.pi * k * n / nfft + phase))).astype(np.complex64)
