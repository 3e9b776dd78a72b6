"""CPU: host-side logic of the package that needs no GPU — frequency axis,
sharding arithmetic, window/argument validation, the numpy mirror of the device
generator."""
import numpy as np
import pytest

from oracle import cpu_ref
from sdr_iq_visualizer_amd import sharding, spectrum, synth


@pytest.mark.parametrize("n,fs,fc", [
    (4096, 1_000_000, 2_400_000_000),       # reference defaults, streamer.py:8-10
    (4096, 1e6, 2.4e9), (65536, 61_440_000, 915_000_000), (1 << 20, 61.44e6, 0.0),
    (2, 1.0, 0.0), (7, 48_000, 100.5), (1001, 2_048_000, 433_920_000), (1, 10.0, 5.0),
])
def test_freq_axis_bit_identical_to_reference_expression(n, fs, fc):
    got = spectrum.freq_axis(n, fs, fc)
    ref = cpu_ref.freq_axis(n, fs, fc)   # np.fft.fftshift(np.fft.fftfreq(n, 1/fs)) + fc
    assert got.dtype == np.float64 and got.shape == (n,)
    assert np.array_equal(got, ref)


def test_freq_axis_cache_hands_out_fresh_arrays():
    """The axis is cached per (n, fs, fc) but every call returns its own array, as the reference's
    expression does: a consumer that edits its copy must not change anybody else's; many distinct keys stay exact."""
    a = spectrum.freq_axis(4096, 1_000_000, 2_400_000_000)
    a[:] = 0.0
    b = spectrum.freq_axis(4096, 1_000_000, 2_400_000_000)
    assert b is not a and np.array_equal(b, cpu_ref.freq_axis(4096, 1_000_000, 2_400_000_000))
    for k in range(20):                                   # more keys than the cache keeps
        fs, fc = 1e6 + 7 * k, 1e9 + 13 * k
        assert np.array_equal(spectrum.freq_axis(1000 + k, fs, fc), cpu_ref.freq_axis(1000 + k, fs, fc))
    assert np.array_equal(spectrum.freq_axis(4096, 1_000_000, 2_400_000_000), b)


def test_freq_axis_rejects_bad_n():
    with pytest.raises(ValueError):
        spectrum.freq_axis(0, 1e6, 0.0)


def test_shard_ranges_cover_and_balance():
    for n in (0, 1, 5, 8, 1000, (1 << 23)):
        for s in (1, 2, 3, 4, 8):
            r = sharding.shard_ranges(n, s)
            assert len(r) == s and r[0][0] == 0 and r[-1][1] == n
            assert all(a[1] == b[0] for a, b in zip(r, r[1:]))
            sizes = [b - a for a, b in r]
            assert max(sizes) - min(sizes) <= 1 and sizes == sorted(sizes, reverse=True)
    assert sharding.shard_ranges(1 << 23, 8)[3] == (3 << 20, 4 << 20)   # BASELINE config 4
    assert sharding.rank_range(10, 1, 4) == (3, 6)
    with pytest.raises(ValueError):
        sharding.shard_ranges(4, 0)


def test_window_spec():
    k, w, key = spectrum._window_spec(None, 8)
    assert (k, w, key) == (0, None, "rect")
    assert spectrum._window_spec("hann", 8)[0] == 1 and spectrum._window_spec("Hanning", 8)[2] == "hann"
    k, w, key = spectrum._window_spec(np.hanning(8), 8)
    assert k == 2 and w.dtype == np.float32 and w.flags.c_contiguous
    with pytest.raises(ValueError):
        spectrum._window_spec(np.ones(7), 8)
    with pytest.raises(ValueError):
        spectrum._window_spec("kaiser", 8)


def test_plan_rejects_bad_nfft_before_touching_the_device():
    for bad in (0, 1, -4, (1 << 21) + 1, (1 << 22) + 2, 1 << 23):
        with pytest.raises(ValueError):
            spectrum.SpectrumPlan(bad)


def test_synth_known_values_and_shape():
    x = synth.synth_iq(1234, 0, 2, 4096)
    assert x.dtype == np.complex64 and x.shape == (2, 4096)
    assert np.all(x.real == np.round(x.real)) and np.all(x.imag == np.round(x.imag))
    assert x.real.min() >= -2048 and x.real.max() <= 2047
    # fmix32 test vectors (MurmurHash3 finaliser)
    assert int(synth.fmix32(np.uint64(0))) == 0
    assert int(synth.fmix32(np.uint64(1))) == 0x514E28B7
    assert int(synth.fmix32(np.uint64(0xFFFFFFFF))) == 0x81F16F39
    # frame numbering is 64-bit: frames beyond 2^32 differ from their low-word aliases
    a = synth.synth_iq(1, (1 << 32) + 5, 1, 64)
    b = synth.synth_iq(1, 5, 1, 64)
    assert not np.array_equal(a, b)
    # a slice of a batch equals the same frames generated alone
    assert np.array_equal(synth.synth_iq(9, 100, 4, 256)[2:], synth.synth_iq(9, 102, 2, 256))
    # roughly zero-mean, full 12-bit spread
    big = synth.synth_iq(7, 0, 16, 4096)
    assert abs(big.real.mean()) < 10 and 1100 < big.real.std() < 1250


def test_tone_is_on_bin():
    t = synth.tone(64, 5.0)
    X = np.fft.fft(t.astype(np.complex128))
    assert int(np.argmax(np.abs(X))) == 5 and abs(abs(X[5]) - 64) < 1e-3


# ---- pin="auto": the decision is arithmetic on measured rates (hostmem.plan_pinning), testable without a GPU ------------

def test_pin_decision_arithmetic():
    import math
    from sdr_iq_visualizer_amd import hostmem as h
    gib = 1 << 30
    # nothing pageable: nothing to decide
    assert h.plan_pinning(8, 8 * gib, 0.0, 1, 16).mode == "as-is"
    # with a few copy threads a single call never pays for its own page-locking (87 ms per GiB against 160 / threads) ...
    for nd in (1, 2, 4, 8):
        for cores in (4, 8, 16, 256):
            d = h.plan_pinning(nd, gib, 1.0, 1, cores)
            assert d.mode == "stage" and d.register_ms > d.staged_ms - d.pinned_ms, (nd, cores)
    # ... on a single core it does: staging 1 GiB costs that core 160 ms, page-locking it 87
    assert h.plan_pinning(2, gib, 1.0, 1, 1).mode == "register"
    # reused buffers: page-locked exactly when the accumulated staging overhead reaches the registration's price
    for nd, cores in ((8, 16), (4, 16), (2, 8), (1, 16)):
        first = h.plan_pinning(nd, nd * gib, 1.0, 1, cores)
        k_star = math.ceil(first.register_ms / (first.staged_ms - first.pinned_ms))
        assert h.plan_pinning(nd, nd * gib, 1.0, k_star - 1, cores).mode == "stage"
        assert h.plan_pinning(nd, nd * gib, 1.0, k_star, cores).mode == "register"
        assert h.plan_pinning(nd, nd * gib, 1.0, k_star + 7, cores).mode == "register"
    # 8 GPUs, 8 GiB, the 16 cores of the GPU box: the host's staging copies bound the call -> warn once;
    # one GPU is link-bound and silent; a small batch is silent; half-pinned calls halve the host's share
    d8 = h.plan_pinning(8, 8 * gib, 1.0, 1, 16)
    assert d8.host_bound and d8.warn and d8.staged_ms > 4 * d8.pinned_ms
    d1 = h.plan_pinning(1, gib, 1.0, 1, 16)
    assert not d1.host_bound and not d1.warn
    assert not h.plan_pinning(8, gib // 2, 1.0, 1, 16).warn
    assert h.plan_pinning(8, 8 * gib, 0.5, 1, 16).staged_ms == pytest.approx(d8.staged_ms / 2)
    # more usable cores than copy threads change nothing; fewer make it worse
    assert h.plan_pinning(8, 8 * gib, 1.0, 1, 64).staged_ms == d8.staged_ms
    assert h.plan_pinning(8, 8 * gib, 1.0, 1, 4).staged_ms == pytest.approx(2 * d8.staged_ms)


def test_auto_pin_counts_sightings_and_registers_once(monkeypatch):
    """The bookkeeping around the decision (no GPU: the registration itself is stubbed)."""
    import warnings
    from sdr_iq_visualizer_amd import hostmem as h
    registered = []
    monkeypatch.setattr(h, "_register_for_life", lambda a: registered.append(a.ctypes.data) or True)
    monkeypatch.setattr(h, "is_pinned", lambda a: a.ctypes.data in registered)
    monkeypatch.setattr(h, "_sightings", {})
    monkeypatch.setattr(h, "_warned", False)
    x = np.zeros((64, 4096), dtype=np.complex64)
    out = np.zeros((64, 4096), dtype=np.float32)
    claimed = 8 << 30                                         # reckon as if the batch were 8 GiB (the arrays stay small)
    modes = []
    with warnings.catch_warnings(record=True) as w:
        warnings.simplefilter("always")
        for _ in range(7):
            modes.append(h.auto_pin([x, out], 8, claimed, cores=16).mode)
    assert modes == ["stage"] * 4 + ["register", "as-is", "as-is"]
    assert sorted(registered) == sorted([x.ctypes.data, out.ctypes.data])
    assert len([m for m in w if issubclass(m.category, ResourceWarning) and "pinned_empty" in str(m.message)]) == 1
    # a view of somebody else's memory cannot carry the finaliser: it stays staged
    monkeypatch.undo()
    assert h._register_for_life(np.frombuffer(bytearray(4096), dtype=np.float32)) is False
    # ... and a small window of a large allocation is not a reason to page-lock all of it
    assert h._register_for_life(np.zeros(1 << 16, dtype=np.float32)[:1024]) is False
