"""Spectrum-row features computed next to the rows on the GPU.

The reference's first consumer of every ``power_db`` row is its rule-based classifier
(app/processing/classifier.py:30-161), which starts by measuring the row with a handful
of O(N) helpers:

    _estimate_noise_floor   :179-181   20th percentile
    _occupied_bandwidth     :163-170   span of bins within 3 / 10 / 20 dB of the peak
    _spectral_flatness      :183-189   geometric / arithmetic mean of linear power
    _spectral_kurtosis      :191-198   fourth standardised moment of the dB values
    _find_peaks             :200-212   strict local maxima above a threshold, greedy spacing
    _peak_spacing_std       :214-219

``row_features`` returns those measurements from two device kernels
(csrc/row_features.hip) — reductions, an exact radix select for the percentile's order
statistics and the peak scan — plus a few scalar combinations done here with the same
dtype rules numpy applies in the reference (float32 percentile interpolation, NEP-50
scalar promotion in the adaptive threshold).  The label ladder (:69-122) and the 12-frame
smoothing (:125-139) are application logic and are not reproduced.

``frame_features`` transforms IQ frames and reduces the rows without the rows ever leaving
the device.
"""
from __future__ import annotations

import ctypes
from typing import List, Optional

import numpy as np

from . import _ffi
from ._ffi import c_void_p, check, lib

_STATS = 16


def _percentile_from_order_stats(n: int, q: float, lo_val: np.ndarray, hi_val: np.ndarray):
    """numpy.percentile(float32 row, q) given sorted[lo], sorted[lo+1]: numpy forms the
    quantile, the virtual index and the interpolation weight in the array's dtype."""
    q32 = np.float32(q) / np.float32(100)
    vi = np.float32(n - 1) * q32
    lo = np.floor(vi)
    gamma = np.float32(vi - lo)
    a, b = lo_val.astype(np.float32), hi_val.astype(np.float32)
    diff = b - a
    out = a + diff * gamma
    if gamma >= np.float32(0.5):
        out = b - diff * (np.float32(1) - gamma)
    return out.astype(np.float32)


def percentile_rank(n: int, q: float) -> int:
    return int(np.floor(np.float32(n - 1) * (np.float32(q) / np.float32(100))))


def percentile_gamma(n: int, q: float) -> np.float32:
    """numpy.percentile's float32 interpolation weight for the q-th percentile of n values."""
    vi = np.float32(n - 1) * (np.float32(q) / np.float32(100))
    return np.float32(vi - np.floor(vi))


def adaptive_threshold(max_db: np.float32, noise_floor_db32: np.float32) -> float:
    """classifier.py:46,55 — ``max(noise_floor + 5, max - 0.9*snr + 5)`` with the dtype rules numpy applies in the
    reference (float32 row statistics, NEP-50 weak python scalars).  The device computes the same value next to
    the rows (row_features_core.h); tests assert the two agree to the bit."""
    noise_floor_db = float(noise_floor_db32)                               # :181
    snr_db = float(max_db - np.float32(noise_floor_db))                    # :46 (float32 - weak python float)
    second = (max_db - np.float32(0.9 * snr_db)) + np.float32(5.0)         # :55, float32 under NEP 50
    first = noise_floor_db + 5.0
    # python max(first, second): `second > first` is evaluated in float32 (first is a weak python float)
    return float(second) if second > np.float32(first) else first


def _assemble(stats, thr, idx, cnt, nfft: int, freqs, percentile: float, max_peaks: int) -> List[dict]:
    """Per-row dicts from the packed device results."""
    n_rows = stats.shape[0]
    mx32 = stats[:, 0].astype(np.float32)
    nf32 = _percentile_from_order_stats(nfft, percentile, stats[:, 1], stats[:, 2])
    f = None if freqs is None else np.asarray(freqs, dtype=np.float64)
    out = []
    for r in range(n_rows):
        s = stats[r]
        peaks = idx[r, : min(int(cnt[r]), max_peaks)].copy()
        sigma = float(np.sqrt(s[4]))
        d = {
            "max_db": float(mx32[r]),
            "argmax": int(s[14]),
            "noise_floor_db": float(nf32[r]),
            "snr_db": float(mx32[r] - nf32[r]),                                        # :46
            "spectral_flatness": float(np.clip(np.exp(s[6]) / s[7], 0.0, 1.0)),      # :186-189
            "spectral_kurtosis": 0.0 if sigma < 1e-9 else float(s[5] / (s[4] * s[4])),  # :195-198
            "adaptive_threshold_db": float(thr[r]),                                   # :55, formed on the device
            "peak_idx": peaks,
            "peak_count": int(cnt[r]),
            "occupied_bins_3db": (int(s[8]), int(s[9])),
            "occupied_bins_10db": (int(s[10]), int(s[11])),
            "occupied_bins_20db": (int(s[12]), int(s[13])),
        }
        if f is not None:
            # an all-NaN row has no bin >= max - x: the kernel returns its empty-range sentinels (first > last)
            # and the reference's _occupied_bandwidth returns 0.0 for it (classifier.py:166-168)
            def _bw(lo, hi):
                lo, hi = int(lo), int(hi)
                return float(f[hi] - f[lo]) if 0 <= lo <= hi < nfft else 0.0
            d["bandwidth_hz_3db"] = _bw(s[8], s[9])                                   # :169-170
            d["bandwidth_hz_10db"] = _bw(s[10], s[11])
            d["bandwidth_hz_20db"] = _bw(s[12], s[13])
            d["peak_spacing_std_hz"] = float(np.std(np.diff(f[peaks]))) if len(peaks) >= 3 else 0.0  # :214-219
            d["peak_density"] = d["peak_count"] / max(nfft, 1)
        out.append(d)
    return out


def _plane_arrays(n_rows: int, max_peaks: int):
    """The caller-side memory of the ``*_planes`` entry points: SDRK_FEAT_PLANES x n_rows 8-byte words + the peak table."""
    return np.empty((_ffi.FEAT_PLANES, n_rows), dtype=np.float64), np.empty((n_rows, max_peaks), dtype=np.int32)


def _arrays_from_planes(planes: np.ndarray, idx: np.ndarray, with_freqs: bool) -> dict:
    """One dict of arrays (entry r = row r) over the planes the library filled (include/sdrk.h, SDRK_FEAT_*): views, no
    per-row Python objects and no arithmetic — the per-row finals of classifier.py:45-58 are formed on the device
    (feature_finalize_kernel).  Same keys as the per-row dicts of ``_assemble``; ``peak_idx`` is ``(rows, max_peaks)``,
    row r holding min(peak_count[r], max_peaks) indices, then -1."""
    n_rows = planes.shape[1]
    ints = planes.view(np.int64)
    out = {
        "max_db": planes[_ffi.FEAT_MAX_DB],
        "argmax": ints[_ffi.FEAT_ARGMAX],
        "noise_floor_db": planes[_ffi.FEAT_NOISE_FLOOR_DB],
        "snr_db": planes[_ffi.FEAT_SNR_DB],                                  # :46
        "spectral_flatness": planes[_ffi.FEAT_FLATNESS],                     # :186-189
        "spectral_kurtosis": planes[_ffi.FEAT_KURTOSIS],                     # :195-198
        "adaptive_threshold_db": planes[_ffi.FEAT_THRESHOLD_DB],             # :55
        "peak_count": ints[_ffi.FEAT_PEAK_COUNT],
        "peak_idx": idx,
    }
    for j, name in enumerate(("3db", "10db", "20db")):
        p = _ffi.FEAT_OCCUPIED_BINS + 2 * j
        out[f"occupied_bins_{name}"] = ints[p:p + 2].reshape(n_rows, 2)
    if with_freqs:
        for j, name in enumerate(("3db", "10db", "20db")):
            out[f"bandwidth_hz_{name}"] = planes[_ffi.FEAT_BANDWIDTH_HZ + j]  # :169-170
        out["peak_spacing_std_hz"] = planes[_ffi.FEAT_PEAK_SPACING_STD_HZ]    # :214-219
        out["peak_density"] = planes[_ffi.FEAT_PEAK_DENSITY]
    return out


def _result_arrays(n_rows: int, max_peaks: int):
    return (np.empty((n_rows, _STATS), dtype=np.float64), np.empty(n_rows, dtype=np.float64),
            np.empty((n_rows, max_peaks), dtype=np.int32), np.empty(n_rows, dtype=np.int32))


def row_features(power_db, freqs=None, *, device: int = 0, percentile: float = 20.0, max_peaks: int = 4096,
                 as_arrays: bool = False):
    """Features of one ``power_db`` row ``(N,)`` -> dict, or of rows ``(R, N)`` -> list of dicts.  One kernel launch
    that reads each row once (``sdrk_row_features``); ``power_db`` may be a host array or ``(device_pointer, n_rows,
    nfft)`` for rows already in HBM.  ``as_arrays=True``: one dict of arrays (entry r = row r) instead of a dict per
    row."""
    _ffi.require_device(device)
    if isinstance(power_db, tuple):
        ptr, n_rows, nfft = c_void_p(int(power_db[0])), int(power_db[1]), int(power_db[2])
        on_device, one = True, False
    else:
        rows = np.ascontiguousarray(np.asarray(power_db, dtype=np.float32))
        one = rows.ndim == 1
        if one:
            rows = rows.reshape(1, -1)
        if rows.ndim != 2 or rows.shape[1] < 1:
            raise ValueError(f"expected (N,) or (R, N) rows, got {rows.shape}")
        ptr, (n_rows, nfft), on_device = rows.ctypes.data_as(c_void_p), rows.shape, False
    if as_arrays:
        planes, idx = _plane_arrays(n_rows, max_peaks)
        f = None if freqs is None else np.ascontiguousarray(freqs, dtype=np.float64)
        if f is not None and f.shape != (nfft,):
            raise ValueError(f"freqs must hold one frequency per bin ({nfft}), got {f.shape}")
        check(lib().sdrk_row_features_planes(device, ptr, int(on_device), ctypes.c_size_t(n_rows), nfft,
                                             percentile_rank(nfft, percentile),
                                             ctypes.c_float(float(percentile_gamma(nfft, percentile))), max(3, nfft // 300),
                                             max_peaks, None if f is None else f.ctypes.data_as(c_void_p),
                                             planes.ctypes.data_as(c_void_p), idx.ctypes.data_as(c_void_p)))
        return _arrays_from_planes(planes, idx, f is not None)
    stats, thr, idx, cnt = _result_arrays(n_rows, max_peaks)
    check(lib().sdrk_row_features(device, ptr, int(on_device), ctypes.c_size_t(n_rows), nfft,
                                  percentile_rank(nfft, percentile), ctypes.c_float(float(percentile_gamma(nfft, percentile))),
                                  max(3, nfft // 300), max_peaks,                                   # :56
                                  stats.ctypes.data_as(c_void_p), thr.ctypes.data_as(c_void_p),
                                  idx.ctypes.data_as(c_void_p), cnt.ctypes.data_as(c_void_p)))
    res = _assemble(stats, thr, idx, cnt, nfft, freqs, percentile, max_peaks)
    return res[0] if one else res


def frame_features(samples, sample_rate: float, center_freq: float, *, window=None, eps: float = 1e-12,
                   device: int = 0, percentile: float = 20.0, max_peaks: int = 4096, return_rows: bool = False,
                   as_arrays: bool = False):
    """IQ frame(s) -> features without the rows leaving the device (streamer.py:119,121 then classifier.py:163-212).
    For 4096-sample frames — the reference's buffer size — the reductions are the epilogue of the transform kernel
    itself and the row exists only on chip; other lengths run the transform and one single-read reduction launch.
    ``return_rows=True`` also returns the ``power_db`` rows (then they are written once and copied back);
    ``as_arrays=True`` returns one dict of arrays instead of a dict per frame (large batches; pass a small ``max_peaks``,
    the peak table is ``(frames, max_peaks)``)."""
    from .spectrum import _as_c64, _cached_plan, freq_axis
    x = _as_c64(samples)
    one = x.ndim == 1
    if one:
        x = x.reshape(1, -1)
    n_rows, nfft = x.shape
    plan = _cached_plan(nfft, window, eps, True, device)
    freqs = freq_axis(nfft, sample_rate, center_freq)
    rows = np.empty((n_rows, nfft), dtype=np.float32) if return_rows else None
    if as_arrays:
        planes, idx = _plane_arrays(n_rows, max_peaks)
        with plan._lock:
            check(lib().sdrk_frame_features_host_planes(plan.handle, x.ctypes.data_as(c_void_p), ctypes.c_size_t(n_rows),
                                                        ctypes.c_size_t(nfft), percentile_rank(nfft, percentile),
                                                        ctypes.c_float(float(percentile_gamma(nfft, percentile))),
                                                        max(3, nfft // 300), max_peaks, freqs.ctypes.data_as(c_void_p),
                                                        planes.ctypes.data_as(c_void_p), idx.ctypes.data_as(c_void_p),
                                                        rows.ctypes.data_as(c_void_p) if rows is not None else None))
        res = _arrays_from_planes(planes, idx, True)
        return (res, rows[0] if one else rows) if return_rows else res     # rows of ONE frame come back as (N,), as in the dict form
    stats, thr, idx, cnt = _result_arrays(n_rows, max_peaks)
    with plan._lock:
        check(lib().sdrk_frame_features_host(plan.handle, x.ctypes.data_as(c_void_p), ctypes.c_size_t(n_rows),
                                             ctypes.c_size_t(nfft), percentile_rank(nfft, percentile),
                                             ctypes.c_float(float(percentile_gamma(nfft, percentile))),
                                             max(3, nfft // 300), max_peaks,
                                             stats.ctypes.data_as(c_void_p), thr.ctypes.data_as(c_void_p),
                                             idx.ctypes.data_as(c_void_p), cnt.ctypes.data_as(c_void_p),
                                             rows.ctypes.data_as(c_void_p) if rows is not None else None))
    res = _assemble(stats, thr, idx, cnt, nfft, freqs, percentile, max_peaks)
    res = res[0] if one else res
    if return_rows:
        return res, (rows[0] if one else rows)
    return res
