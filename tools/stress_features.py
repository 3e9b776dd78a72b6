"""Developer helper (GPU box): randomised rows through the per-row reductions (classifier.py:163-219 restated in
oracle/cpu_ref.py): lengths on both sides of the register-resident limit, value distributions that exercise the
histogram select (smooth noise), its fall-backs (heavy ties, far-off percentiles, -inf values) and the peak scan
(dense and sparse candidates, spacings from the rule max(3, n // 300)).  python tools/stress_features.py [cases] [seed]"""
import sys
import numpy as np
sys.path.insert(0, __file__.rsplit("/", 2)[0])
from oracle import cpu_ref
from sdr_iq_visualizer_amd import features

kinds = ["noise", "quantised", "tone", "steps", "wide", "holes", "const_tail", "ramp", "huge", "nans"]


def run(cases, seed):
    rng = np.random.default_rng(seed)
    done = {k: 0 for k in kinds}
    for c in range(cases):
        one_case(c, rng, done)
    return done


def one_case(c, rng, done):
    if True:
        n = int(rng.choice([64, 100, 257, 1000, 2048, 4000, 4096, 4097, 8192, 30000, 40000, 70000, 131072]))
        kind = kinds[c % len(kinds)]
        x = (rng.standard_normal(n) * rng.uniform(1, 9) + rng.uniform(-120, 60)).astype(np.float32)
        if kind == "quantised":                       # heavy ties: more than 64 equal values around the percentile
            x = (np.round(x * 2) / 2).astype(np.float32)
        elif kind == "tone":                          # a few bins far above a floor: percentile 60+ dB under the mean? no: under the max
            x[rng.integers(0, n, 3)] += np.float32(90)
        elif kind == "steps":                         # two plateaus: the 20th percentile sits on a cliff
            x[: n // 5 + int(rng.integers(-3, 4))] -= np.float32(150)
        elif kind == "wide":                          # 80 % of the bins 70 dB over the rest: percentile far below the mean
            x[rng.random(n) < 0.8] += np.float32(70)
        elif kind == "holes":                         # eps = 0 rows: -inf bins
            x[rng.random(n) < 0.05] = -np.inf
        elif kind == "const_tail":
            x[n // 2:] = x[0]
        elif kind == "huge":                          # past float32's 10^(x/10) range (385 dB) in part of the row
            x[rng.random(n) < 0.3] += np.float32(rng.uniform(400, 3000))
        elif kind == "nans":                          # a few NaN bins: the reference's max / percentile turn NaN
            x[rng.integers(0, n, int(rng.integers(1, 4)))] = np.nan
        elif kind == "ramp":
            x = np.linspace(-80, 5, n).astype(np.float32) + (rng.standard_normal(n) * 0.01).astype(np.float32)
        freqs = cpu_ref.freq_axis(n, 2e6, 1e9)
        got, ref = features.row_features(x, freqs), cpu_ref.row_features(freqs, x)
        for key in ("noise_floor_db", "snr_db", "bandwidth_hz_3db", "bandwidth_hz_10db", "bandwidth_hz_20db", "adaptive_threshold_db",
                    "peak_spacing_std_hz"):
            a, b = got[key], ref[key]
            assert a == b or (np.isnan(a) and np.isnan(b)), (c, kind, n, key, a, b)
        assert np.array_equal(got["peak_idx"], ref["peak_idx"]), (c, kind, n)
        if kind != "nans":                            # (np.argmax points at a NaN; the device's extra field skips NaNs)
            assert got["argmax"] == int(np.argmax(x)), (c, kind, n)
        for key, tol in (("spectral_flatness", 1e-6), ("spectral_kurtosis", 1e-9)):
            a, b = got[key], ref[key]
            assert (np.isnan(a) and np.isnan(b)) or abs(a - b) <= tol * max(1.0, abs(b)), (c, kind, n, key, a, b)
        # the array form of the same row (sdrk_row_features_planes: finals formed on the device)
        arr = features.row_features(x[None, :], freqs, as_arrays=True, max_peaks=4096)
        for key in ("noise_floor_db", "snr_db", "bandwidth_hz_3db", "bandwidth_hz_10db", "bandwidth_hz_20db", "adaptive_threshold_db",
                    "spectral_kurtosis", "max_db", "argmax", "peak_count"):
            a, b = arr[key][0], got[key]
            assert a == b or (np.isnan(a) and np.isnan(b)), (c, kind, n, key, a, b, "array form")
        a, b = arr["spectral_flatness"][0], got["spectral_flatness"]
        assert (np.isnan(a) and np.isnan(b)) or abs(a - b) <= 1e-14, (c, kind, n, "flatness", a, b)
        a, b = arr["peak_spacing_std_hz"][0], got["peak_spacing_std_hz"]
        assert abs(a - b) <= 1e-9 * max(1.0, abs(b)), (c, kind, n, "spacing", a, b)
        k = min(int(arr["peak_count"][0]), 4096)
        assert np.array_equal(arr["peak_idx"][0][:k], got["peak_idx"][:k]), (c, kind, n, "array form peaks")
        done[kind] += 1


if __name__ == "__main__":
    print("all ok:", run(int(sys.argv[1]) if len(sys.argv) > 1 else 200, int(sys.argv[2]) if len(sys.argv) > 2 else 1))
