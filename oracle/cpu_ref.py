"""CPU oracle — TEST INFRASTRUCTURE, NOT PRODUCT CODE.

A numpy restatement of the reference's streamed-IQ spectrum path, used only as
the checker in tests/, in ``__graft_entry__.smoke()`` and as the ``cpu_baseline``
leg of ``bench.py``.  Nothing under ``sdr-iq-visualizer_amd/`` imports it.

Each function cites the reference lines it follows (paths relative to the
reference checkout).  The arithmetic itself lives in numpy (``numpy.fft`` =
pocketfft; the reference pins numpy==2.3.2 at requirements.txt:47, this image
has numpy 2.2.x — both compute complex64 input in single precision).

Pinning: the reference's own tests hold NO golden vectors for this path
(tests/test_streamer.py patches threading.Thread, so lines
119-121 never run).  This oracle is therefore pinned against outputs of the
reference itself, captured by ``oracle/make_golden.py`` (which imports
``app.sdr.streamer`` with ``adi`` mocked, as the reference's own tests do, and
drives ``_stream_data``) into ``tests/golden/*.npz``;
``tests/test_oracle_golden.py`` checks this file against them bit for bit.
"""
from __future__ import annotations

from collections import deque

import numpy as np


def spectrum_db(frames, window=None, eps=1e-12, shift=True):
    """``power_db`` of app/sdr/streamer.py:119,121, batched over the last axis.

    Expression order is the reference's: fft -> fftshift -> abs -> + eps -> log10
    -> * 20.  ``window`` (absent in the reference = rectangular) multiplies the
    samples first, in the input precision.  complex64 in -> float32 out,
    complex128 in -> float64 out, exactly as numpy does for the reference.
    """
    x = np.asarray(frames)
    if window is not None:
        w = np.asarray(window)
        if x.dtype == np.complex64:
            w = w.astype(np.float32)
        x = x * w
    fft_data = np.fft.fft(x, axis=-1)                       # streamer.py:119
    if shift:
        fft_data = np.fft.fftshift(fft_data, axes=-1)       # streamer.py:119
    return 20 * np.log10(np.abs(fft_data) + eps)            # streamer.py:121


def fft(frames, window=None, shift=False):
    """``np.fft.fft`` of streamer.py:119 alone (optionally shifted), same dtype rules."""
    x = np.asarray(frames)
    if window is not None:
        w = np.asarray(window)
        if x.dtype == np.complex64:
            w = w.astype(np.float32)
        x = x * w
    X = np.fft.fft(x, axis=-1)
    return np.fft.fftshift(X, axes=-1) if shift else X


def freq_axis(n, sample_rate, center_freq):
    """``freqs`` of app/sdr/streamer.py:120."""
    return np.fft.fftshift(np.fft.fftfreq(n, 1 / sample_rate)) + center_freq


def hann(n):
    """Symmetric Hann = numpy.hanning(n): the window matplotlib's psd() applies in
    scripts/process_sigmf_data.py:188 (mlab.window_hanning)."""
    return np.hanning(n)


def stft_db(iq, nfft, hop, window=None, eps=1e-12, shift=True):
    """Row r = spectrum_db(iq[r*hop : r*hop+nfft]); rows = 1 + (L - nfft)//hop."""
    iq = np.asarray(iq).reshape(-1)
    rows = 0 if iq.shape[0] < nfft else 1 + (iq.shape[0] - nfft) // hop
    out_dtype = np.float32 if iq.dtype == np.complex64 else np.float64
    out = np.empty((rows, nfft), dtype=out_dtype)
    for r in range(rows):
        out[r] = spectrum_db(iq[r * hop: r * hop + nfft], window=window, eps=eps, shift=shift)
    return out


def welch_psd(iq, nfft, sample_rate, hop=None, window=None, shift=True):
    """matplotlib ``mlab.psd(x, NFFT, Fs, window=hanning, noverlap=NFFT-hop)`` for complex
    input (two-sided, no detrend, scale_by_freq): the quantity the reference's offline script
    plots at scripts/process_sigmf_data.py:188-189 (``plt.psd`` = 10*log10 of it).
    mean over the 1+(L-NFFT)//hop full segments of |fft(w*seg)|^2 / (Fs * sum(w^2)),
    rolled so that DC sits at index NFFT/2."""
    x = np.asarray(iq).reshape(-1)
    hop = nfft if hop is None else hop
    w = np.hanning(nfft) if window is None else np.asarray(window, dtype=np.float64)
    rows = 1 + (x.shape[0] - nfft) // hop
    acc = np.zeros(nfft, dtype=np.float64)
    for r in range(rows):
        seg = x[r * hop: r * hop + nfft].astype(np.complex128) * w
        acc += np.abs(np.fft.fft(seg)) ** 2
    pxx = acc / rows / (sample_rate * np.sum(w ** 2))
    return np.fft.fftshift(pxx) if shift else pxx


def row_features(freqs, power_db):
    """The measurements classify_signal_advanced makes before its rule ladder
    (app/processing/classifier.py:45-60), restated helper by helper (:163-219)."""
    x = np.asarray(power_db)
    n = len(x)
    noise_floor_db = float(np.percentile(x, 20))                              # :179-181
    snr_db = float(np.max(x) - noise_floor_db)                                # :46

    def obw(drop):                                                            # :163-170
        mask = x >= np.max(x) - float(drop)
        occ = np.asarray(freqs)[mask]
        return float(occ[-1] - occ[0]) if np.any(mask) else 0.0

    p = np.clip(np.power(10.0, np.asarray(x, dtype=float) / 10.0), 1e-15, None)   # :184-186
    sfm = float(np.clip(float(np.exp(np.mean(np.log(p)))) / float(np.mean(p)), 0.0, 1.0))
    xf = np.asarray(x, dtype=float)                                           # :191-198
    mu, sigma = float(np.mean(xf)), float(np.std(xf))
    kurt = 0.0 if sigma < 1e-9 else float(np.mean(((xf - mu) / sigma) ** 4))
    thr = max(noise_floor_db + 5.0, np.max(x) - 0.9 * snr_db + 5.0)           # :55
    min_dist = max(3, n // 300)                                               # :56
    peaks, last = [], -min_dist                                               # :200-212
    for i in range(1, n - 1):
        if xf[i] > thr and xf[i] > xf[i - 1] and xf[i] > xf[i + 1] and i - last >= min_dist:
            peaks.append(i)
            last = i
    spacing = float(np.std(np.diff(np.asarray(freqs)[peaks]))) if len(peaks) >= 3 else 0.0   # :214-219
    return {"noise_floor_db": noise_floor_db, "snr_db": snr_db, "bandwidth_hz_3db": obw(3),
            "bandwidth_hz_10db": obw(10), "bandwidth_hz_20db": obw(20), "spectral_flatness": sfm,
            "spectral_kurtosis": kurt, "adaptive_threshold_db": float(thr), "peak_idx": np.array(peaks, dtype=np.int64),
            "peak_count": len(peaks), "peak_spacing_std_hz": spacing}


class Waterfall:
    """app/dashboard/callbacks.py:19,176,182: ``deque(maxlen=100)``, ``append(power_db)``,
    ``np.array(deque)`` — rows oldest first; y axis = range(len) (:186)."""

    def __init__(self, maxlen=100):
        self.rows = deque(maxlen=maxlen)                     # callbacks.py:19

    def append(self, power_db):
        self.rows.append(power_db)                           # callbacks.py:176

    def __len__(self):
        return len(self.rows)

    def as_array(self):
        return np.array(self.rows)                           # callbacks.py:182


def dashboard_peak_markers(power_db):
    """The indices update_graphs marks on the spectrum trace (app/dashboard/callbacks.py:148-159): scipy's find_peaks
    with distance max(5, N//200) and prominence 3, or — without scipy — strict local maxima above median + 5 dB."""
    power_db = np.asarray(power_db)
    try:
        from scipy.signal import find_peaks                       # callbacks.py:150
        dist = max(5, len(power_db) // 200)                      # :152
        idx, _ = find_peaks(power_db, distance=dist, prominence=3)   # :153
        return np.asarray(idx, dtype=np.int64)
    except ImportError:
        med = float(np.median(power_db))                          # :156-159
        return np.array([i for i in range(1, len(power_db) - 1)
                         if power_db[i] > power_db[i - 1] and power_db[i] > power_db[i + 1] and power_db[i] > med + 5],
                        dtype=np.int64)


def process_frame(samples, sample_rate, center_freq, now=None):
    """The ``plot_data`` dict of app/sdr/streamer.py:119-130."""
    import time
    return {
        "time": time.time() if now is None else now,
        "samples": samples,
        "freqs": freq_axis(len(samples), sample_rate, center_freq),
        "power_db": spectrum_db(samples),
        "sample_rate": sample_rate,
        "center_freq": center_freq,
    }
