"""CPU: the "next" rows of SURVEY.md §8f that are host logic — SigMF cf32_le I/O (f3) and
the reader-loop shim's queue / error behaviour (f2, with the oracle injected as the per-frame
transform: there is no GPU here) — plus the Welch restatement pinned to matplotlib's mlab.psd."""
import io
import json
import time
import zipfile

import numpy as np
import pytest

from oracle import cpu_ref
from sdr_iq_visualizer_amd import sigmf_io, streaming, synth


def test_oracle_welch_matches_mlab_psd(golden):
    g = golden["ref_welch"]
    fs = float(g["fs"][0])
    p = cpu_ref.welch_psd(g["iq"], 1024, fs)
    assert np.abs(p - g["pxx"]).max() <= 1e-12 * g["pxx"].max()
    p2 = cpu_ref.welch_psd(g["iq"], 1024, fs, hop=512)
    assert np.abs(p2 - g["pxx_noverlap512"]).max() <= 1e-12 * g["pxx"].max()
    assert np.array_equal(cpu_ref.freq_axis(1024, fs, 0.0), g["freqs"])


def test_sigmf_roundtrip_and_metadata_keys(tmp_path):
    x = synth.synth_iq(3, 0, 2, 4096).reshape(-1)
    data, meta = sigmf_io.write_sigmf(str(tmp_path / "rec"), x, 1_000_000, 2_400_000_000)
    assert data.endswith(".sigmf-data") and meta.endswith(".sigmf-meta")
    m = json.load(open(meta))
    # keys the reference's exporter writes (app/dashboard/callbacks.py:285-304)
    assert m["global"]["core:datatype"] == "cf32_le" and m["global"]["core:sample_rate"] == 1_000_000
    assert m["global"]["core:version"] == "1.0.0" and m["annotations"] == []
    assert m["captures"][0]["core:frequency"] == 2_400_000_000 and m["captures"][0]["core:sample_start"] == 0
    raw = np.fromfile(data, dtype="<f4")
    assert raw.size == 2 * x.size and raw[0] == x[0].real and raw[1] == x[0].imag     # interleaved I,Q
    for path in (meta, data, str(tmp_path / "rec")):
        y, info = sigmf_io.read_sigmf(path)
        assert y.dtype == np.complex64 and np.array_equal(y, x)
        assert info["sample_rate"] == 1_000_000.0 and info["center_freq"] == 2_400_000_000.0
    y, _ = sigmf_io.read_sigmf(meta, max_samples=100)
    assert np.array_equal(y, x[:100])


def test_sigmf_zip_as_the_dashboard_exports_it(tmp_path):
    x = synth.synth_iq(4, 0, 1, 4096)[0].astype(np.complex128)      # the app holds complex128 samples
    blob = sigmf_io.to_zip_bytes(x, 1_000_000, 2_400_000_000)
    with zipfile.ZipFile(io.BytesIO(blob)) as z:
        names = sorted(z.namelist())
    assert names == ["sdr_sample.sigmf-data", "sdr_sample.sigmf-meta"]
    p = tmp_path / "dl.zip"
    p.write_bytes(blob)
    y, info = sigmf_io.read_sigmf(str(p))
    assert np.array_equal(y, x.astype(np.complex64)) and info["center_freq"] == 2.4e9


def test_sigmf_integer_datatypes_and_errors(tmp_path):
    base = str(tmp_path / "i16")
    iq = np.array([1, -2, 3, -4, 5, -6], dtype="<i2")
    iq.tofile(base + ".sigmf-data")
    json.dump({"global": {"core:datatype": "ci16_le", "core:sample_rate": 48000}, "captures": [{}]},
              open(base + ".sigmf-meta", "w"))
    y, info = sigmf_io.read_sigmf(base)
    assert np.array_equal(y, np.array([1 - 2j, 3 - 4j, 5 - 6j], dtype=np.complex64)) and info["center_freq"] == 0.0
    json.dump({"global": {"core:datatype": "rf32_le"}}, open(base + ".sigmf-meta", "w"))
    with pytest.raises(ValueError):
        sigmf_io.read_sigmf(base)


def _oracle_compute(samples, fs, fc):
    return cpu_ref.process_frame(samples, fs, fc)


def test_streamer_shim_fifo_and_drop_oldest():
    src = streaming.SyntheticSource(nfft=256, seed=5, tone_bin=None)
    s = streaming.SpectrumStreamer(src, 2_000_000, 915_000_000, queue_size=4, compute=_oracle_compute)
    assert s.is_connected() and s.get_latest_data() is None
    for i in range(7):                                   # drive the loop body by hand: 7 pushes into 4 slots
        s._push({"i": i})
    assert [s.get_latest_data()["i"] for _ in range(4)] == [3, 4, 5, 6]       # oldest dropped, FIFO pop
    assert s.get_latest_data() is None
    assert s.start_streaming()
    deadline = time.time() + 5
    while s.total_frames < 6 and time.time() < deadline:
        time.sleep(0.01)
    s.stop_streaming()
    st = s.get_status()
    assert st["total_frames"] >= 6 and st["queue_size"] == 4 and not st["running"]
    d = s.get_latest_data()
    assert list(d) == ["time", "samples", "freqs", "power_db", "sample_rate", "center_freq"]
    assert d["power_db"].shape == (256,) and d["freqs"][128] == 915_000_000.0


def test_streamer_shim_stops_after_three_consecutive_errors(monkeypatch):
    class Broken:
        calls = 0

        def rx(self):
            Broken.calls += 1
            raise OSError(110, "timed out")

    monkeypatch.setattr(streaming.time, "sleep", lambda s: None)
    s = streaming.SpectrumStreamer(Broken(), compute=_oracle_compute)
    s.running = True
    s._stream_data()                                     # returns by itself
    # errno 110 asks for a reconnect; this source has no reconnect hook, so it is marked disconnected and the
    # next iteration gives up (the reference does the same when its reconnects fail: fixture scenario
    # "unreachable_errno113_reconnect_fails_then_backoff")
    assert Broken.calls == 1 and not s.running and not s.connected and s.total_frames == 0

    class Glitchy:
        calls = 0

        def rx(self):
            Glitchy.calls += 1
            raise ValueError("no frame")

    s = streaming.SpectrumStreamer(Glitchy(), compute=_oracle_compute)
    s.running = True
    s._stream_data()
    assert Glitchy.calls == 3 and not s.running and s.total_frames == 0   # third plain error in a row: stop

    class Flaky:                                         # one failure, then fine: counter resets
        n = 0

        def rx(self):
            Flaky.n += 1
            if Flaky.n in (2, 4):
                raise ValueError("glitch")
            if Flaky.n >= 8:
                s2.running = False
            return synth.synth_iq(1, Flaky.n, 1, 64)[0]

    s2 = streaming.SpectrumStreamer(Flaky(), compute=_oracle_compute)
    s2.running = True
    s2._stream_data()
    assert s2.total_frames == 6                          # 8 reads, 2 failed, never 3 in a row


def _replay(scenario):
    """Run one scripted scenario of tests/golden/ref_streamer_failures.json through SpectrumStreamer and return
    its event trace in the fixture's format."""
    trace = []
    errors = {"Exception": Exception, "OSError": OSError, "ValueError": ValueError, "RuntimeError": RuntimeError}
    outcomes = list(scenario["reconnect"])
    frame = np.ones(64, dtype=np.complex64)

    class Source:
        i = 0

        def rx(self):
            if self.i >= len(scenario["rx"]):
                s.running = False                        # script used up: one last good frame ends the loop
                return frame
            step = scenario["rx"][self.i]
            self.i += 1
            if step == "ok":
                return frame
            if "return" in step:
                return None                              # the transform rejects it
            cls = errors[step["raise"]]
            raise cls(step["errno"], "scripted failure") if "errno" in step else cls("scripted failure")

        def reconnect(self):
            ok = outcomes.pop(0) if outcomes else False
            trace.append(["reconnect", ok])
            return ok

    def compute(samples, fs, fc):
        if samples is None:
            raise TypeError("not a frame")
        return cpu_ref.process_frame(samples, fs, fc)

    s = streaming.SpectrumStreamer(Source(), compute=compute)
    s.connected = bool(scenario.get("start_connected", True))
    s._sleep = lambda d: trace.append(["sleep", round(float(d), 6)])
    push = s._push
    s._push = lambda data: (trace.append(["frame"]), push(data))[1]
    s.running = True
    s._stream_data()
    return {"events": trace, "running": bool(s.running), "connected": bool(s.connected),
            "total_frames": int(s.total_frames)}


def test_streamer_shim_failure_behaviour_matches_reference_traces():
    """f2: every wait, reconnect and frame of the reference's own loop (app/sdr/streamer.py:83-174) under scripted
    failures — back-off sequence, errno classes, reconnect schedules, stop-vs-resume decisions — reproduced
    event for event."""
    import json
    import os
    from tests.conftest import GOLDEN
    fixture = json.load(open(os.path.join(GOLDEN, "ref_streamer_failures.json")))
    assert len(fixture["scenarios"]) >= 15
    for sc in fixture["scenarios"]:
        assert _replay(sc) == sc["trace"], sc["name"]


def test_streamer_shim_equals_reference_loop_on_random_failure_scripts():
    """Where the reference is present (the build container): 300 random scripts — good frames, generic exceptions,
    OSErrors of every errno class the loop distinguishes (9, 10054, 110, 113, others, none), frames the transform
    rejects, reconnects that succeed on the k-th try or never, starting connected or not — through the reference's
    own `_stream_data` (driven as oracle/make_golden_streamer.py drives it) and through SpectrumStreamer: the same
    waits, reconnects and frames in the same order, and the same final state."""
    import logging
    import os
    import sys
    from unittest.mock import MagicMock
    ref_root = os.environ.get("SDRK_REFERENCE", "/root/reference")
    if not os.path.isdir(os.path.join(ref_root, "app", "sdr")):
        pytest.skip("the reference is not present on this machine")
    from oracle import make_golden_streamer as G
    sys.modules.setdefault("adi", MagicMock())
    sys.path.insert(0, ref_root)
    try:
        import app.sdr.streamer as module
    finally:
        while ref_root in sys.path:
            sys.path.remove(ref_root)
    rng = np.random.default_rng(83)
    steps = ["ok", {"raise": "Exception"}, {"raise": "ValueError"}, {"raise": "RuntimeError"}, {"raise": "OSError"},
             {"raise": "OSError", "errno": 9}, {"raise": "OSError", "errno": 10054}, {"raise": "OSError", "errno": 110},
             {"raise": "OSError", "errno": 113}, {"raise": "OSError", "errno": 5}, {"return": "garbage"}]
    weights = np.array([6, 2, 1, 1, 1, 1, 1, 2, 1, 1, 1], dtype=float)
    logging.disable(logging.CRITICAL)
    try:
        for c in range(300):
            sc = {"name": f"random_{c}",
                  "rx": [steps[i] for i in rng.choice(len(steps), size=int(rng.integers(1, 14)), p=weights / weights.sum())],
                  "reconnect": [bool(b) for b in rng.random(int(rng.integers(0, 6))) < 0.4],
                  "start_connected": bool(rng.random() < 0.85)}
            want = G.run_scenario(module, module.SDRDataStreamer, sc)
            assert _replay(sc) == want, sc
    finally:
        logging.disable(logging.NOTSET)


def test_sigmf_source_cuts_and_loops(tmp_path):
    x = synth.synth_iq(8, 0, 3, 64).reshape(-1)
    sigmf_io.write_sigmf(str(tmp_path / "r"), x, 1e6, 0)
    src = streaming.SigMFSource(str(tmp_path / "r.sigmf-meta"), nfft=64)
    frames = [src.rx() for _ in range(4)]
    assert np.array_equal(frames[0], x[:64]) and np.array_equal(frames[2], x[128:]) and np.array_equal(frames[3], x[:64])
    src2 = streaming.SigMFSource(str(tmp_path / "r"), nfft=64, loop=False)
    for _ in range(3):
        src2.rx()
    with pytest.raises(EOFError):
        src2.rx()


def test_oracle_row_features_match_reference_classifier_helpers(golden):
    """f1: the oracle's restatement of app/processing/classifier.py:163-219 against values the
    reference's own helpers returned (oracle/make_golden.py)."""
    g = golden["ref_classifier_features"]
    for k in (str(n) for n in g["names"]):
        f = cpu_ref.row_features(g["freqs"], g[f"{k}/power_db"])
        got = np.array([f["noise_floor_db"], f["snr_db"], f["bandwidth_hz_3db"], f["bandwidth_hz_10db"],
                        f["bandwidth_hz_20db"], f["spectral_flatness"], f["spectral_kurtosis"],
                        f["adaptive_threshold_db"], f["peak_spacing_std_hz"]])
        assert np.array_equal(got, g[f"{k}/scalars"]), k
        assert np.array_equal(f["peak_idx"], g[f"{k}/peak_idx"]), k


def test_oracle_row_features_match_reference_on_corner_rows(golden):
    """The same restatement where realistic rows never go: 41 rows chosen for the corners of the device reductions
    (lengths 1 ... 33000, ties, cliffs at the percentile, -inf bins, an all-NaN row, the all-zero frame's constant
    row, overflowing and fully clipped flatness), against what the reference's helpers returned for them
    (oracle/make_golden_corner_rows.py).  Bit for bit, NaN where the reference gives NaN."""
    import warnings
    g = golden["ref_classifier_corner_rows"]
    fs, fc = g["fs_fc"]
    for k in (str(n) for n in g["names"]):
        p = g[f"{k}/power_db"]
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            f = cpu_ref.row_features(cpu_ref.freq_axis(p.shape[0], fs, fc), p)
        got = np.array([f["noise_floor_db"], f["snr_db"], f["bandwidth_hz_3db"], f["bandwidth_hz_10db"],
                        f["bandwidth_hz_20db"], f["spectral_flatness"], f["spectral_kurtosis"],
                        f["adaptive_threshold_db"], f["peak_spacing_std_hz"]])
        assert np.array_equal(got, g[f"{k}/scalars"], equal_nan=True), (k, got, g[f"{k}/scalars"])
        assert np.array_equal(f["peak_idx"], g[f"{k}/peak_idx"]), k


def test_oracle_equals_reference_helpers_on_random_rows():
    """Where the reference is present (the build container; it never travels), the oracle's row_features is run next to
    the reference's own helpers on 400 random rows of the kinds tools/stress_features.py and stress_fused.py use
    (noise, ties, tones, cliffs, far-off percentiles, -inf bins, constant tails, ramps, values past 385 dB; lengths
    16 ... 20000): every scalar and every peak list identical.  This is what lets the GPU stress tools use the oracle
    as their checker for thousands of rows."""
    import os
    import sys
    import warnings
    ref_root = os.environ.get("SDRK_REFERENCE", "/root/reference")
    if not os.path.isdir(os.path.join(ref_root, "app", "processing")):
        pytest.skip("the reference is not present on this machine")
    sys.path.insert(0, ref_root)
    try:
        from app.processing import classifier as C
    finally:
        sys.path.remove(ref_root)
    rng = np.random.default_rng(400)
    kinds = ["noise", "quantised", "tone", "steps", "wide", "holes", "const_tail", "ramp", "huge", "nans"]
    for c in range(400):
        n = int(rng.choice([16, 64, 100, 257, 1000, 2048, 4096, 4097, 8192, 20000]))
        kind = kinds[c % len(kinds)]
        x = (rng.standard_normal(n) * rng.uniform(1, 9) + rng.uniform(-120, 60)).astype(np.float32)
        if kind == "quantised":
            x = (np.round(x * 2) / 2).astype(np.float32)
        elif kind == "tone":
            x[rng.integers(0, n, 3)] += np.float32(90)
        elif kind == "steps":
            x[: max(1, n // 5 + int(rng.integers(-3, 4)))] -= np.float32(150)
        elif kind == "wide":
            x[rng.random(n) < 0.8] += np.float32(70)
        elif kind == "holes":
            x[rng.random(n) < 0.05] = -np.inf
        elif kind == "const_tail":
            x[n // 2:] = x[0]
        elif kind == "ramp":
            x = np.linspace(-80, 5, n).astype(np.float32) + (rng.standard_normal(n) * 0.01).astype(np.float32)
        elif kind == "huge":
            x[rng.random(n) < 0.3] += np.float32(rng.uniform(400, 3000))
        elif kind == "nans":
            x[rng.integers(0, n, int(rng.integers(1, 4)))] = np.nan
        freqs = cpu_ref.freq_axis(n, 20e6, 2.4e9)
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            nf = C._estimate_noise_floor(x)
            snr = float(np.max(x) - nf)
            thr = max(nf + 5.0, np.max(x) - 0.9 * snr + 5.0)
            pk = C._find_peaks(x, threshold_db=thr, min_distance_bins=max(3, len(x) // 300))
            ref = np.array([nf, snr, C._occupied_bandwidth(freqs, x, 3), C._occupied_bandwidth(freqs, x, 10),
                            C._occupied_bandwidth(freqs, x, 20), C._spectral_flatness(x), C._spectral_kurtosis(x), float(thr),
                            C._peak_spacing_std(freqs, pk)])
            f = cpu_ref.row_features(freqs, x)
        got = np.array([f["noise_floor_db"], f["snr_db"], f["bandwidth_hz_3db"], f["bandwidth_hz_10db"], f["bandwidth_hz_20db"],
                        f["spectral_flatness"], f["spectral_kurtosis"], f["adaptive_threshold_db"], f["peak_spacing_std_hz"]])
        assert np.array_equal(got, ref, equal_nan=True), (c, kind, n, got, ref)
        assert np.array_equal(f["peak_idx"], np.array(pk, dtype=np.int64)), (c, kind, n)


def test_oracle_welch_equals_mlab_psd_for_random_segmentations():
    """scripts/process_sigmf_data.py:188-189 plots matplotlib's psd (mlab.psd, Hann, two-sided): the oracle's
    welch_psd next to mlab.psd itself for random segment lengths (odd ones too), overlaps and rates, to 1e-12 of the
    largest bin.  Skipped where matplotlib is absent; the committed ref_welch.npz is what travels."""
    mlab = pytest.importorskip("matplotlib.mlab")
    rng = np.random.default_rng(3)
    for c in range(40):
        n = int(rng.choice([16, 64, 100, 256, 1000, 1024, 4096]))
        hop = int(rng.integers(1, n + 1))
        segs = int(rng.integers(1, 30))
        L = n + (segs - 1) * hop + int(rng.integers(0, hop))
        fs = float(rng.choice([1e6, 2.4e6, 61.44e6]))
        x = ((rng.standard_normal(L) + 1j * rng.standard_normal(L)) * rng.uniform(0.01, 100)).astype(np.complex64)
        pxx, _ = mlab.psd(x, NFFT=n, Fs=fs, window=mlab.window_hanning, noverlap=n - hop)
        got = cpu_ref.welch_psd(x, n, fs, hop=hop)
        assert np.abs(got - pxx).max() <= 1e-12 * pxx.max(), (c, n, hop, segs)


def test_freq_axis_equals_numpy_expression_for_random_lengths_and_rates():
    """streamer.py:120, `fftshift(fftfreq(N, 1/fs)) + fc`: the host axis equals numpy's expression bit for bit for
    odd and even lengths, powers of two up to 2^20, and arbitrary rates and centre frequencies."""
    import sdr_iq_visualizer_amd as pkg
    rng = np.random.default_rng(1)
    for t in range(600):
        n = int(rng.integers(1, 5000)) if t % 3 else int(2 ** rng.integers(1, 21))
        fs = float(rng.choice([1e6, 2.4e6, 61.44e6, 20e6, rng.uniform(1e3, 1e9)]))
        fc = float(rng.choice([0.0, 2.4e9, 100e6, rng.uniform(0, 6e9)]))
        assert np.array_equal(pkg.freq_axis(n, fs, fc), np.fft.fftshift(np.fft.fftfreq(n, 1 / fs)) + fc), (n, fs, fc)


def test_percentile_rank_and_interpolation_mirror_numpy():
    from sdr_iq_visualizer_amd import features
    rng = np.random.default_rng(9)
    for n in (5, 64, 300, 1000, 4096, 65536):
        x = rng.standard_normal(n).astype(np.float32)
        s = np.sort(x)
        for q in (20.0, 50.0, 95.0, 33.3):
            r = features.percentile_rank(n, q)
            hi = min(r + 1, n - 1)
            got = features._percentile_from_order_stats(n, q, np.array([s[r]]), np.array([s[hi]]))[0]
            assert got == np.percentile(x, q), (n, q)


# ---- f2: the dashboard callback, fed by the shim -----------------------------------------------

def _oracle_streamer(radio):
    from oracle.make_golden_dashboard import CENTER_FREQ, SAMPLE_RATE
    return streaming.SpectrumStreamer(radio, SAMPLE_RATE, CENTER_FREQ, compute=cpu_ref.process_frame)


def _exact_row(got, ref, what):
    assert np.array_equal(got, ref), what


def test_dashboard_record_replays_through_the_shim_and_the_deque(golden):
    """Host logic, no reference needed: the streamer shim (oracle transform injected: no GPU here) and the oracle's
    deque reproduce every call the reference's update_graphs made — which frame it drew, its peak markers, which rows
    the heatmap held (queue drop-oldest over 105 frames, deque wrap after 100 rows), x frozen, y = range(rows)."""
    from tests import dashboard_replay
    g = golden["ref_update_graphs"]
    seen = {}
    for sc in dashboard_replay.scenarios(g):
        seen[sc["name"]] = dashboard_replay.replay(g, sc, _oracle_streamer, lambda nfft: cpu_ref.Waterfall(100), _exact_row)
    assert seen == {"live4096": 12, "wrap512": 107}


def _reference_callback():
    import os
    ref = os.environ.get("SDRK_REFERENCE", "/root/reference")
    if not os.path.isdir(os.path.join(ref, "app", "dashboard")):
        pytest.skip("the reference checkout is not here (build container only)")
    pytest.importorskip("dash")
    pytest.importorskip("plotly")
    from oracle import make_golden_dashboard as mk
    import logging
    logging.disable(logging.CRITICAL)
    try:
        cb, update_graphs = mk.load_callbacks()
    finally:
        logging.disable(logging.NOTSET)
    return mk, cb, update_graphs


def test_reference_update_graphs_runs_unchanged_on_the_shim(golden):
    """The reference's OWN callback (app/dashboard/callbacks.py:95-243, imported here, `adi` mocked), once fed by the
    reference's streamer and once by sdr_iq_visualizer_amd.streaming.SpectrumStreamer: the same calls draw the same
    frames, the four figures of every call serialise to the same JSON, and both equal the committed fixture."""
    mk, cb, update_graphs = _reference_callback()
    g = golden["ref_update_graphs"]
    for sc in mk.SCENARIOS:
        ref_rec, ref_digests = mk.capture(cb, update_graphs, sc)
        shim_rec, shim_digests = mk.capture(cb, update_graphs, sc, make_streamer=_oracle_streamer)
        assert ref_digests == shim_digests and len(ref_digests) > 10
        assert ref_rec["calls"] == shim_rec["calls"]
        assert np.array_equal(ref_rec["power_db"], shim_rec["power_db"])
        # ... and the committed record is what the reference produces today
        k = sc["name"]
        assert np.array_equal(g[f"{k}/power_db"], ref_rec["power_db"])
        assert np.array_equal(g[f"{k}/call_frame"], [c["frame"] for c in ref_rec["calls"]])
        assert np.array_equal(g[f"{k}/peaks_concat"], [v for c in ref_rec["calls"] for v in c["peaks"]])
        assert np.array_equal(g[f"{k}/z_ids_concat"], [v for c in ref_rec["calls"] for v in c["z_ids"]])
        assert np.array_equal(g[f"{k}/trace_x"], ref_rec["trace_x"]) and np.array_equal(g[f"{k}/heat_x"], ref_rec["heat_x"])
        assert json.loads(str(g[f"{k}/call_text_json"])) == [[c["status"], c["cls"]] for c in ref_rec["calls"]]
