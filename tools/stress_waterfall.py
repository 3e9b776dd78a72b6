"""Developer helper (GPU box): a random sequence of operations on a WaterfallBuffer against ``collections.deque`` — the
reference's own container (app/dashboard/callbacks.py:19,176,182): rows appended from the host (must come back bit for
bit), IQ appended as packed frames, as an overlapped stream and from a device buffer (synchronously and only
enqueued), clear, full and partial read-outs, bin-decimated read-outs (max-hold / mean) in one and in two phases.
python tools/stress_waterfall.py [sequences] [seed]"""
import ctypes
import sys
from collections import deque

import numpy as np

sys.path.insert(0, __file__.rsplit("/", 2)[0])
from oracle import cpu_ref
import sdr_iq_visualizer_amd as pkg
from sdr_iq_visualizer_amd import _ffi
from tests.parity import assert_db_parity


def _check_rows(got, ref_rows, exact, what):
    assert got.shape == (len(ref_rows), got.shape[1]) and got.dtype == np.float32, (what, got.shape, len(ref_rows))
    for r, (row, ex) in enumerate(zip(ref_rows, exact)):
        if ex:
            assert np.array_equal(got[r], row), f"{what}: host-appended row {r} changed"
        else:
            assert_db_parity(got[r], row, what=f"{what} row {r}")


def one_sequence(rng, steps):
    nfft = int(rng.choice([16, 64, 256, 1000, 4096, 8192, 32768, 1 << 20], p=[0.135] * 6 + [0.13, 0.06]))
    maxlen = int(rng.choice([1, 2, 3, 7, 100]))
    if nfft == 1 << 20:          # the by-16 max-hold companion rows beside the ring (round 5): few, long rows
        maxlen, steps = int(rng.choice([1, 2, 3])), min(steps, 14)
    window = None if rng.random() < 0.5 else "hann"
    eps = float(rng.choice([1e-12, 1e-10]))
    w = cpu_ref.hann(nfft) if window else None
    ref, exact = deque(maxlen=maxlen), deque(maxlen=maxlen)
    lib = _ffi.lib()
    counts = {}

    def iq(samples):
        return ((rng.standard_normal(samples) + 1j * rng.standard_normal(samples)) * rng.uniform(0.1, 500)).astype(np.complex64)

    def push(rows, ex):
        for row in rows:
            ref.append(np.asarray(row, dtype=np.float32))
            exact.append(ex)

    with pkg.WaterfallBuffer(nfft, maxlen, window=window, eps=eps) as wf:
        for _ in range(steps):
            op = str(rng.choice(["rows", "iq", "stream", "device", "device_async", "clear", "read", "read_part", "decimated", "two_phase"],
                                p=[0.2, 0.12, 0.1, 0.08, 0.08, 0.04, 0.14, 0.08, 0.1, 0.06]))
            counts[op] = counts.get(op, 0) + 1
            what = f"N={nfft} maxlen={maxlen} window={window} op={op}"
            if op == "rows":
                k = int(rng.integers(1, 2 * maxlen + 4)) if maxlen < 50 else int(rng.integers(1, 130))
                rows = (rng.standard_normal((k, nfft)) * 20 - 60).astype(np.float32)
                wf.append(rows if k > 1 or rng.random() < 0.5 else rows[0])
                push(rows, True)
            elif op == "iq":
                k = int(rng.integers(1, 6))
                x = iq(k * nfft).reshape(k, nfft)
                wf.append(x if k > 1 else x[0])
                push(cpu_ref.spectrum_db(x, window=w, eps=eps), False)
            elif op == "stream":
                hop = int(rng.integers(1, 2 * nfft + 1))
                k = int(rng.integers(0, 5))
                L = nfft + (k - 1) * hop + int(rng.integers(0, hop)) if k else int(rng.integers(0, nfft))
                x = iq(L)
                wf.append_iq(x, hop=hop)
                if k:
                    push(cpu_ref.stft_db(x, nfft, hop, window=w, eps=eps), False)
            elif op in ("device", "device_async"):
                k = int(rng.integers(1, 5))
                x = iq(k * nfft)
                d = ctypes.c_void_p()
                _ffi.check(lib.sdrk_dev_alloc(0, x.nbytes, ctypes.byref(d)))
                _ffi.check(lib.sdrk_memcpy_h2d(0, d, x.ctypes.data_as(ctypes.c_void_p), x.nbytes))
                wf.append_iq_device(d.value, k, wait=(op == "device"))
                if op == "device_async" and rng.random() < 0.5:
                    wf.sync()
                else:
                    assert len(wf) == min(maxlen, len(ref) + k), what          # the count is known before the work ends
                    wf.sync()                                                   # (the buffer is freed below)
                _ffi.check(lib.sdrk_dev_free(0, d))
                push(cpu_ref.spectrum_db(x.reshape(k, nfft), window=w, eps=eps), False)
            elif op == "clear":
                wf.clear()
                ref.clear(); exact.clear()
            elif op == "read":
                _check_rows(wf.as_array(), list(ref), list(exact), what)
            elif op == "read_part":
                m = int(rng.integers(0, maxlen + 3))
                got = wf.as_array(max_rows=m)
                keep = min(m, len(ref))
                _check_rows(got, list(ref)[len(ref) - keep:], list(exact)[len(ref) - keep:], what)
            elif op in ("decimated", "two_phase"):
                divisors = [f for f in (2, 4, 5, 8, 16, 64, 250, 1024) if nfft % f == 0 and f <= nfft]
                f = int(rng.choice(divisors))
                mode = str(rng.choice(["max", "mean"]))
                m = None if rng.random() < 0.5 else int(rng.integers(1, maxlen + 2))
                if op == "decimated":
                    got = wf.as_array(max_rows=m, decimate=f, mode=mode)
                else:
                    wf.gather_begin(m, decimate=f, mode=mode)
                    got = wf.gather_end()
                keep = len(ref) if m is None else min(m, len(ref))
                rows = np.array(list(ref)[len(ref) - keep:], dtype=np.float32).reshape(keep, nfft // f, f)
                want = rows.max(axis=2) if mode == "max" else rows.astype(np.float64).mean(axis=2)
                assert got.shape == (keep, nfft // f), (what, got.shape, keep)
                # host-appended rows decimate exactly (max) / to float32 rounding (mean); transformed rows carry the
                # transform's own tolerance (1e-5 of the frame peak: up to ~0.01 dB on a bin 40 dB down) — what this
                # checks is which rows and which bins are combined, and an error there shows as whole dB
                tol = 0.05 if not all(list(exact)[len(ref) - keep:]) else (0.0 if mode == "max" else 1e-4)
                assert np.all(np.abs(got.astype(np.float64) - want) <= tol), (what, f, mode, float(np.abs(got - want).max()))
            assert len(wf) == len(ref), (what, len(wf), len(ref))
        _check_rows(wf.as_array(), list(ref), list(exact), f"N={nfft} maxlen={maxlen} final")
    return counts


def run(sequences, seed, steps=40):
    rng = np.random.default_rng(seed)
    total = {}
    for _ in range(sequences):
        for k, v in one_sequence(rng, steps).items():
            total[k] = total.get(k, 0) + v
    return total


if __name__ == "__main__":
    print("all ok:", run(int(sys.argv[1]) if len(sys.argv) > 1 else 30, int(sys.argv[2]) if len(sys.argv) > 2 else 1))
