"""Developer helper (GPU box): IQ frames of many kinds through ``features.frame_features`` — at N = 4096 the reductions
are the epilogue of the transform kernel (fft4096_features.hip), at other lengths the transform plus the single-read
kernel — against the oracle's restatement of the classifier helpers (classifier.py:163-219) evaluated on the rows the
device returned.  The kinds aim at the reductions' corners: silence (every bin -240.00002 dB: one value 4096 times),
an impulse (a flat row), pure tones on and off a bin (rows 100+ dB deep), bursts, 12-bit integers, clipped input,
a DC offset, two tones 60 dB apart, and frames of a batch that differ in kind.  python tools/stress_fused.py [cases] [seed]"""
import sys
import numpy as np
sys.path.insert(0, __file__.rsplit("/", 2)[0])
from oracle import cpu_ref
from sdr_iq_visualizer_amd import features
from tests.parity import assert_db_parity

KINDS = ["noise", "silence", "impulse", "tone_on_bin", "tone_off_bin", "burst", "adc12", "clipped", "dc", "two_tones", "tiny"]


def frame(rng, kind, n):
    t = np.arange(n)
    noise = (rng.standard_normal(n) + 1j * rng.standard_normal(n))
    if kind == "noise":
        x = noise * rng.uniform(0.01, 1000)
    elif kind == "silence":
        x = np.zeros(n, dtype=np.complex64)
    elif kind == "impulse":
        x = np.zeros(n, dtype=np.complex64)
        x[int(rng.integers(0, n))] = rng.uniform(0.5, 50)
    elif kind == "tone_on_bin":
        x = np.exp(2j * np.pi * int(rng.integers(-n // 2, n // 2)) * t / n) * rng.uniform(0.1, 100)
    elif kind == "tone_off_bin":
        x = np.exp(2j * np.pi * rng.uniform(-n / 2, n / 2) * t / n) * rng.uniform(0.1, 100) + noise * 1e-3
    elif kind == "burst":
        x = noise * 30
        a = int(rng.integers(0, n))
        x[a: a + int(rng.integers(1, n))] = 0
    elif kind == "adc12":
        x = np.round(noise * 300).clip(-2048, 2047)
    elif kind == "clipped":
        x = (noise * 3000)
        x = x.real.clip(-2048, 2047) + 1j * x.imag.clip(-2048, 2047)
    elif kind == "dc":
        x = noise + rng.uniform(10, 1000)
    elif kind == "two_tones":
        x = np.exp(2j * np.pi * 0.11 * t) * 100 + np.exp(2j * np.pi * -0.23 * t) * 0.1 + noise * 1e-3
    else:  # tiny: magnitudes near the additive floor of streamer.py:121
        x = noise * 1e-11
    return np.asarray(x, dtype=np.complex64)


def run(cases, seed):
    rng = np.random.default_rng(seed)
    done = {k: 0 for k in KINDS}
    for c in range(cases):
        n = 4096 if rng.random() < 0.7 else int(rng.choice([256, 1024, 2048, 8192, 16384]))
        b = int(rng.integers(1, 9))
        window = None if rng.random() < 0.5 else "hann"
        kinds = [KINDS[(c + i * int(rng.integers(0, 2))) % len(KINDS)] for i in range(b)]
        x = np.stack([frame(rng, k, n) for k in kinds])
        fs, fc = 2e6, 1e9
        got, rows = features.frame_features(x, fs, fc, window=window, return_rows=True)
        blind = features.frame_features(x, fs, fc, window=window)
        assert_db_parity(rows, cpu_ref.spectrum_db(x, window=cpu_ref.hann(n) if window else None), what=f"case {c} rows")
        freqs = cpu_ref.freq_axis(n, fs, fc)
        # the array form (per-row finals formed on the device): every quantity against the per-frame dicts
        mp = int(rng.choice([8, 64, 1024]))
        arr = features.frame_features(x, fs, fc, window=window, as_arrays=True, max_peaks=mp)
        capped = features.frame_features(x, fs, fc, window=window, max_peaks=mp)
        for r in range(b):
            for key in ("max_db", "argmax", "noise_floor_db", "snr_db", "spectral_kurtosis", "adaptive_threshold_db", "peak_count",
                        "bandwidth_hz_3db", "bandwidth_hz_10db", "bandwidth_hz_20db", "peak_density"):
                a, e = arr[key][r], capped[r][key]
                assert a == e or (np.isnan(a) and np.isnan(e)), ((c, r, kinds[r], n, window), key, a, e, "array form")
            a, e = arr["spectral_flatness"][r], capped[r]["spectral_flatness"]
            assert (np.isnan(a) and np.isnan(e)) or abs(a - e) <= 1e-14, ((c, r, kinds[r]), "flatness", a, e)
            a, e = arr["peak_spacing_std_hz"][r], capped[r]["peak_spacing_std_hz"]
            assert abs(a - e) <= 1e-9 * max(1.0, abs(e)), ((c, r, kinds[r]), "spacing", a, e)
            k = min(int(arr["peak_count"][r]), mp)
            assert np.array_equal(arr["peak_idx"][r][:k], capped[r]["peak_idx"]) and np.all(arr["peak_idx"][r][k:] == -1), (c, r)
            for j, name in enumerate(("3db", "10db", "20db")):
                assert tuple(arr[f"occupied_bins_{name}"][r]) == capped[r][f"occupied_bins_{name}"], (c, r, name)
        for r in range(b):
            ref = cpu_ref.row_features(freqs, rows[r])
            tag = (c, r, kinds[r], n, window)
            for key in ("noise_floor_db", "snr_db", "bandwidth_hz_3db", "bandwidth_hz_10db", "bandwidth_hz_20db",
                        "adaptive_threshold_db", "peak_spacing_std_hz"):
                a, e = got[r][key], ref[key]
                assert a == e or (np.isnan(a) and np.isnan(e)), (tag, key, a, e)
                assert blind[r][key] == a or (np.isnan(a) and np.isnan(blind[r][key])), (tag, key, "rows written or not")
            assert np.array_equal(got[r]["peak_idx"], ref["peak_idx"]) and np.array_equal(blind[r]["peak_idx"], ref["peak_idx"]), tag
            assert got[r]["argmax"] == int(np.argmax(rows[r])), tag
            for key, tol in (("spectral_flatness", 1e-6), ("spectral_kurtosis", 1e-9)):
                a, e = got[r][key], ref[key]
                assert (np.isnan(a) and np.isnan(e)) or abs(a - e) <= tol * max(1.0, abs(e)), (tag, key, a, e)
            done[kinds[r]] += 1
    return done


if __name__ == "__main__":
    print("all ok:", run(int(sys.argv[1]) if len(sys.argv) > 1 else 200, int(sys.argv[2]) if len(sys.argv) > 2 else 1))
