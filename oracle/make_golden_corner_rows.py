#!/usr/bin/env python3
"""Generate tests/golden/ref_classifier_corner_rows.npz: the REFERENCE's classifier helpers
(app/processing/classifier.py:163-219, called the way :45-58 calls them) on rows chosen for the corners of the
device reductions — lengths on both sides of every internal limit, ties, cliffs, far-off percentiles, -inf bins,
an all-NaN row, the all-zero frame's constant row.

TEST INFRASTRUCTURE.  Runs only in the build container (it needs /root/reference, numpy only); the committed .npz —
inputs and the reference's outputs, data only — is what travels.  ref_classifier_features.npz (make_golden.py) holds
realistic rows the reference produced itself; this file pins the oracle (oracle/cpu_ref.row_features) and the device
where realistic rows never go.

    python oracle/make_golden_corner_rows.py
"""
from __future__ import annotations

import os
import sys
import warnings

import numpy as np

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = os.environ.get("SDRK_REFERENCE", "/root/reference")
OUT = os.path.join(REPO, "tests", "golden", "ref_classifier_corner_rows.npz")


def corner_rows():
    rng = np.random.default_rng(2024)
    rows = {}

    def noise(n, sigma=5.0, mean=-70.0):
        return (rng.standard_normal(n) * sigma + mean).astype(np.float32)

    for n in (16, 64, 100, 257, 1000, 2048, 4095, 4096, 4097, 8192, 33000):
        rows[f"noise_n{n}"] = noise(n)
    rows["silence_4096"] = np.full(4096, np.float32(-240.00002), dtype=np.float32)       # the all-zero frame's row
    rows["constant_1000"] = np.full(1000, np.float32(-3.5), dtype=np.float32)
    q = noise(4096)
    rows["quantised_half_db_4096"] = (np.round(q * 2) / 2).astype(np.float32)             # > 64 values tied at the percentile
    rows["quantised_half_db_8192"] = (np.round(noise(8192) * 2) / 2).astype(np.float32)
    t = noise(4096, 2.0, -90.0); t[[700, 2048, 3100]] += np.float32(90)
    rows["three_tones_4096"] = t
    s = noise(4096); s[: 4096 // 5 + 2] -= np.float32(150)
    rows["cliff_at_percentile_4096"] = s                                                  # 20th percentile sits on a 150 dB step
    s2 = noise(2048); s2[: 2048 // 5 - 3] -= np.float32(150)
    rows["cliff_below_percentile_2048"] = s2
    w = noise(4096, 4.0, -100.0); w[rng.random(4096) < 0.8] += np.float32(70)
    rows["percentile_far_below_mean_4096"] = w
    w2 = noise(2048, 3.0, -40.0); w2[rng.random(2048) < 0.8] += np.float32(70)
    rows["percentile_far_below_mean_2048"] = w2                                           # (the row behind the contracted-fma finding)
    h = np.linspace(-50, 10, 4096).astype(np.float32); h[::7] = -np.inf
    rows["holes_minus_inf_4096"] = h
    h2 = noise(8192); h2[rng.random(8192) < 0.05] = -np.inf
    rows["holes_minus_inf_8192"] = h2
    c = noise(4096); c[2048:] = c[0]
    rows["constant_tail_4096"] = c
    rows["ramp_4096"] = (np.linspace(-80, 5, 4096) + rng.standard_normal(4096) * 0.01).astype(np.float32)
    rows["ramp_down_1000"] = np.linspace(0, -120, 1000).astype(np.float32)
    p = noise(4096, 0.5, -80.0); p[::13] += np.float32(30)
    rows["comb_every_13_bins_4096"] = p                                                   # spacing == min_distance (4096 // 300 = 13)
    p2 = noise(4096, 0.5, -80.0); p2[::12] += np.float32(30)
    rows["comb_every_12_bins_4096"] = p2                                                  # one bin closer than min_distance
    e = noise(4096, 0.5, -80.0); e[0] = e[-1] = np.float32(0)
    rows["maxima_on_both_edges_4096"] = e                                                 # edge bins are never peaks (:207)
    pl = noise(4096, 0.5, -80.0); pl[1000:1004] = np.float32(-20)
    rows["plateau_is_no_peak_4096"] = pl                                                  # strict local maximum (:207)
    rows["all_nan_64"] = np.full(64, np.nan, dtype=np.float32)
    rows["huge_positive_256"] = noise(256, 5.0, 3000.0)                                   # 10^(x/10) overflows: flatness inf/inf
    rows["very_negative_256"] = noise(256, 5.0, -400.0)                                   # all bins clipped to 1e-15 (:186)
    m = noise(1024, 5.0, -152.0)
    rows["straddles_the_clip_1024"] = m                                                   # some bins under -150 dB, some over
    # (added after the rows above so that their random draws stay what they were)
    pn = noise(4096); pn[[5, 1700, 4000]] = np.nan
    rows["three_nans_in_4096"] = pn                                                       # np.percentile / np.max -> NaN: no peaks, 0 Hz
    pn2 = noise(300); pn2[17] = np.nan
    rows["one_nan_in_300"] = pn2
    pn3 = noise(9000); pn3[8000] = np.nan
    rows["one_nan_in_9000"] = pn3
    for n in (1, 2, 3, 5):                                                                # shorter than one peak neighbourhood
        rows[f"tiny_n{n}"] = noise(n)
    rows["tiny_peak_n3"] = np.array([-80, -20, -80], dtype=np.float32)
    return rows


def main():
    sys.path.insert(0, REF)
    from app.processing import classifier as C  # noqa: E402
    store, names = {}, []
    fs, fc = 20_000_000.0, 2_400_000_000.0
    for name, p in corner_rows().items():
        n = p.shape[0]
        freqs = np.fft.fftshift(np.fft.fftfreq(n, 1 / fs)) + fc                          # streamer.py:120
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            nf = C._estimate_noise_floor(p)                                               # classifier.py:45
            snr = float(np.max(p) - nf)                                                   # :46
            thr = max(nf + 5.0, np.max(p) - 0.9 * snr + 5.0)                              # :55
            pk = C._find_peaks(p, threshold_db=thr, min_distance_bins=max(3, len(p) // 300))   # :56
            scal = np.array([nf, snr, C._occupied_bandwidth(freqs, p, 3), C._occupied_bandwidth(freqs, p, 10),
                             C._occupied_bandwidth(freqs, p, 20), C._spectral_flatness(p), C._spectral_kurtosis(p),
                             float(thr), C._peak_spacing_std(freqs, pk)], dtype=np.float64)
        names.append(name)
        store[f"{name}/power_db"] = p
        store[f"{name}/scalars"] = scal
        store[f"{name}/peak_idx"] = np.array(pk, dtype=np.int64)
    store["names"] = np.array(names)
    store["fs_fc"] = np.array([fs, fc])
    np.savez_compressed(OUT, **store)
    print("wrote", OUT, len(names), "rows,", os.path.getsize(OUT) // 1024, "KiB")


if __name__ == "__main__":
    main()
