#!/usr/bin/env python3
"""Generate tests/golden/*.npz by running the REFERENCE's own code on fixed inputs.

TEST INFRASTRUCTURE.  Runs only in the build container (it needs /root/reference);
the committed .npz files are what travels.  The reference's spectrum path has no
function boundary, so this drives it the way its own tests set it up:
``sys.modules['adi'] = MagicMock()`` (reference tests/test_streamer.py:8-9), then
``SDRDataStreamer._stream_data()`` (app/sdr/streamer.py:95) with a fake ``sdr``
whose ``rx()`` returns the fixed frame once and then stops the loop; the result
is read back with ``get_latest_data()`` (:196).  What is stored is data only:
inputs and the reference's outputs.

    python oracle/make_golden.py            # rewrites tests/golden/
"""
from __future__ import annotations

import os
import sys
from unittest.mock import MagicMock

import numpy as np

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = os.environ.get("SDRK_REFERENCE", "/root/reference")
OUT = os.path.join(REPO, "tests", "golden")

sys.path.insert(0, REPO)
import importlib  # noqa: E402

synth = importlib.import_module("sdr_iq_visualizer_amd.synth")  # input generator only (no GPU use)


def load_reference_streamer():
    if "adi" not in sys.modules:
        sys.modules["adi"] = MagicMock()
    sys.path.insert(0, REF)
    from app.sdr.streamer import SDRDataStreamer  # noqa: E402
    return SDRDataStreamer


class FakeSdr:
    """rx() hands out the queued frames, then stops the reader loop."""

    def __init__(self, streamer, frames):
        self.streamer, self.frames, self.i = streamer, list(frames), 0

    def rx(self):
        frame = self.frames[self.i]
        self.i += 1
        if self.i >= len(self.frames):
            self.streamer.running = False
        return frame


def run_reference(SDRDataStreamer, frames, sample_rate=1_000_000, center_freq=2_400_000_000):
    s = SDRDataStreamer(sample_rate=sample_rate, center_freq=center_freq)
    s.sdr = FakeSdr(s, frames)
    s.connected = True
    s.running = True
    s._stream_data()
    out = []
    while True:
        d = s.get_latest_data()
        if d is None:
            break
        out.append(d)
    assert len(out) == len(frames), (len(out), len(frames))
    return out


def golden_inputs_4096():
    n = 4096
    cases = {}
    cases["pluto12_noise"] = synth.synth_iq(seed=1234, first_frame=0, n_frames=1, nfft=n)[0]
    rng = np.random.default_rng(0)
    cases["gauss_noise"] = (rng.standard_normal(n) + 1j * rng.standard_normal(n)).astype(np.complex64)
    cases["tone_onbin_k100"] = synth.tone(n, 100.0)
    cases["tone_offbin_k100p37"] = synth.tone(n, 100.37)
    cases["zeros"] = np.zeros(n, dtype=np.complex64)
    imp0 = np.zeros(n, dtype=np.complex64); imp0[0] = 1.0
    imp1 = np.zeros(n, dtype=np.complex64); imp1[1] = 1.0
    cases["impulse_n0"] = imp0
    cases["impulse_n1"] = imp1
    cases["dc"] = np.ones(n, dtype=np.complex64)
    cases["two_tones_60db"] = (synth.tone(n, -700.0) + synth.tone(n, 900.0, amplitude=1e-3)).astype(np.complex64)
    cases["tone_plus_pluto_noise"] = (synth.synth_iq(7, 5, 1, n)[0] * np.float32(0.01)
                                      + synth.tone(n, 1234.0, amplitude=500.0)).astype(np.complex64)
    return cases


def main():
    os.makedirs(OUT, exist_ok=True)
    S = load_reference_streamer()

    # ---- N = 4096 (the reference's live frame size, streamer.py:10) ----
    cases = golden_inputs_4096()
    names = sorted(cases)
    frames64 = [cases[k] for k in names]
    ref64 = run_reference(S, frames64)
    ref128 = run_reference(S, [f.astype(np.complex128) for f in frames64])
    store = {"names": np.array(names)}
    for k, d64, d128 in zip(names, ref64, ref128):
        assert d64["power_db"].dtype == np.float32 and d128["power_db"].dtype == np.float64
        store[f"{k}/iq"] = cases[k]
        store[f"{k}/power_db_c64"] = d64["power_db"]          # reference output, c64 input
        store[f"{k}/power_db_c128"] = d128["power_db"]        # reference output, c128 input
    store["freqs_default"] = ref64[0]["freqs"]                # fs=1e6, fc=2.4e9 (streamer.py:8-9)
    np.savez_compressed(os.path.join(OUT, "ref_n4096.npz"), **store)

    # ---- other frame sizes the reference handles through len(samples) ----
    store = {}
    for n in (2, 8, 64, 256, 1024, 2048, 8192):
        x = synth.synth_iq(seed=99, first_frame=n, n_frames=1, nfft=n)[0]
        d = run_reference(S, [x], sample_rate=2_000_000, center_freq=915_000_000)[0]
        store[f"n{n}/iq"] = x
        store[f"n{n}/power_db_c64"] = d["power_db"]
        store[f"n{n}/freqs"] = d["freqs"]
    np.savez_compressed(os.path.join(OUT, "ref_other_sizes.npz"), **store)

    # ---- large frames: seeds + sampled bins only (inputs regenerate from synth) ----
    store = {}
    for n in (65536, 1 << 20):
        x = (synth.synth_iq(seed=4321, first_frame=3, n_frames=1, nfft=n)[0]
             + synth.tone(n, n / 8 + 0.25, amplitude=300.0)).astype(np.complex64)
        d = run_reference(S, [x], sample_rate=61_440_000, center_freq=2_400_000_000)[0]
        p = d["power_db"]
        idx = np.unique(np.concatenate([
            np.linspace(0, n - 1, 48).astype(np.int64),
            np.array([n // 2, n // 2 + n // 8, n // 2 + n // 8 + 1, int(np.argmax(p))])]))
        store[f"n{n}/seed"] = np.array([4321, 3])
        store[f"n{n}/tone_bin_amp"] = np.array([n / 8 + 0.25, 300.0])
        store[f"n{n}/idx"] = idx
        store[f"n{n}/power_db_c64_at_idx"] = p[idx]
        store[f"n{n}/argmax"] = np.array([int(np.argmax(p))])
        store[f"n{n}/sum_db"] = np.array([float(np.sum(p.astype(np.float64)))])
        store[f"n{n}/freqs_ends"] = d["freqs"][[0, n // 2, -1]]
    np.savez_compressed(os.path.join(OUT, "ref_large_sampled.npz"), **store)

    # ---- offline Welch PSD: what plt.psd at scripts/process_sigmf_data.py:188-189 computes ----
    # (matplotlib is a third-party library, not reference code; the script itself needs the
    # absent `sigmf` package, so its psd call is made directly with the script's arguments.)
    from matplotlib import mlab
    x = (synth.synth_iq(seed=11, first_frame=0, n_frames=40, nfft=1024).reshape(-1) * np.float32(1 / 2048)
         + synth.tone(40 * 1024, 40 * 1024 * 0.1037, amplitude=2.0)).astype(np.complex64)
    fs = 2_000_000.0
    pxx, f = mlab.psd(x.astype(np.complex128), NFFT=1024, Fs=fs)          # defaults: hanning, noverlap=0
    pxx_ov, _ = mlab.psd(x.astype(np.complex128), NFFT=1024, Fs=fs, noverlap=512)
    np.savez_compressed(os.path.join(OUT, "ref_welch.npz"), iq=x, fs=np.array([fs]), pxx=pxx, freqs=f,
                        pxx_noverlap512=pxx_ov)

    # ---- classifier feature helpers (app/processing/classifier.py:163-219), run as the reference
    # calls them at :45-58, on rows the reference itself produced above plus an OFDM-like comb ----
    from app.processing import classifier as C
    g4 = np.load(os.path.join(OUT, "ref_n4096.npz"))
    rows = {k: g4[f"{k}/power_db_c64"] for k in ("pluto12_noise", "gauss_noise", "tone_offbin_k100p37",
                                                 "two_tones_60db", "tone_plus_pluto_noise", "impulse_n1")}
    comb = synth.synth_iq(21, 0, 1, 4096)[0] * np.float32(0.02)
    for kk in range(-1500, 1501, 100):
        comb = comb + synth.tone(4096, kk, amplitude=30.0)
    rows["comb_31_tones"] = run_reference(S, [comb.astype(np.complex64)], 20_000_000, 2_400_000_000)[0]["power_db"]
    freqs = run_reference(S, [comb.astype(np.complex64)], 20_000_000, 2_400_000_000)[0]["freqs"]
    store = {"names": np.array(sorted(rows)), "freqs": freqs}
    for k in sorted(rows):
        p = rows[k]
        nf = C._estimate_noise_floor(p)
        snr = float(np.max(p) - nf)
        thr = max(nf + 5.0, np.max(p) - 0.9 * snr + 5.0)
        pk = C._find_peaks(p, threshold_db=thr, min_distance_bins=max(3, len(p) // 300))
        store[f"{k}/power_db"] = p
        store[f"{k}/scalars"] = np.array([nf, snr, C._occupied_bandwidth(freqs, p, 3), C._occupied_bandwidth(freqs, p, 10),
                                          C._occupied_bandwidth(freqs, p, 20), C._spectral_flatness(p),
                                          C._spectral_kurtosis(p), float(thr), C._peak_spacing_std(freqs, pk)], dtype=np.float64)
        store[f"{k}/peak_idx"] = np.array(pk, dtype=np.int64)
    np.savez_compressed(os.path.join(OUT, "ref_classifier_features.npz"), **store)

    # ---- waterfall: the reference's deque semantics (callbacks.py:19,176,182) ----
    # The callback module needs dash/plotly and a live streamer, so the deque lines
    # are exercised directly: 103 appends into deque(maxlen=100) -> rows 3..102.
    from collections import deque
    dq = deque(maxlen=100)
    rows = (np.arange(103, dtype=np.float32)[:, None] + np.linspace(0, 1, 16, dtype=np.float32)[None, :])
    for r in rows:
        dq.append(r)
    np.savez_compressed(os.path.join(OUT, "ref_waterfall.npz"), rows_in=rows, array_out=np.array(dq))
    print("wrote", sorted(os.listdir(OUT)))


if __name__ == "__main__":
    main()
