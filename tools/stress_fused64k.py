#!/usr/bin/env python3
"""Developer helper (GPU box): the persistent N = 65536 launch (fft_fused64k.hip) against the two tiled launches on random
workloads — frames per call, hop (packed, half-overlapped, odd strides), window, eps, shift, first-sample offset — every row of
every call compared bit for bit through a 64-bit checksum computed on the device rows copied back in slabs.  A stale or early read
of the L2-resident ring would corrupt whole 16 x 16 blocks of a row.
    python3 tools/stress_fused64k.py [calls] [seed]          (run(calls, seed) returns the number of calls checked)"""
import ctypes
import sys

import numpy as np

sys.path.insert(0, __file__.rsplit("/", 2)[0])
from sdr_iq_visualizer_amd import _ffi  # noqa: E402
from sdr_iq_visualizer_amd.spectrum import SpectrumPlan  # noqa: E402

N = 65536


def run(calls=40, seed=1, max_frames=2600, verbose=False):
    lib = _ffi.lib()
    rng = np.random.default_rng(seed)
    cap_samples = (max_frames + 2) * N
    d_in, d_a, d_b = ctypes.c_void_p(), ctypes.c_void_p(), ctypes.c_void_p()
    _ffi.check(lib.sdrk_dev_alloc(0, cap_samples * 8, ctypes.byref(d_in)))
    _ffi.check(lib.sdrk_dev_alloc(0, max_frames * N * 4, ctypes.byref(d_a)))
    _ffi.check(lib.sdrk_dev_alloc(0, max_frames * N * 4, ctypes.byref(d_b)))
    slab = 64
    a, b = np.empty((slab, N), np.float32), np.empty((slab, N), np.float32)
    done = 0
    try:
        _ffi.check(lib.sdrk_synth_fill(0, 1000 + seed, 0, cap_samples // 4096, 4096, d_in, None))
        for c in range(calls):
            kind = int(rng.integers(0, 4))
            window = [None, "hann", "hann", rng.random(N).astype(np.float32) + 0.25][kind]
            shift = bool(rng.integers(0, 2))
            eps = float(rng.choice([1e-12, 1e-10, 0.0]))
            hop = int(rng.choice([N, N // 2, N // 2, N // 4, 3 * N // 4, int(rng.integers(1, N)), N + 4096 * int(rng.integers(0, 3))]))
            nf_max = min(max_frames, (cap_samples - N) // hop + 1)
            nf = int(rng.integers(512, max(513, nf_max)))
            off = int(rng.integers(0, max(1, cap_samples - ((nf - 1) * hop + N)))) & ~1           # first sample (8-byte aligned anyway)
            with SpectrumPlan(N, window=window, eps=eps, shift=shift) as pf, \
                    SpectrumPlan(N, window=window, eps=eps, shift=shift, fused64k=False) as pt:
                reps = int(rng.integers(1, 4))
                for _ in range(reps):                                   # back to back: ring slots hot, flags from the last launch around
                    pf.exec_device(d_in.value + off * 8, nf, d_a.value, frame_stride=hop)
                pf.sync()
                st = pf.fused_status()
                assert st["launches"] == reps and not st["fallen_back"], st
                pt.exec_device(d_in.value + off * 8, nf, d_b.value, frame_stride=hop)
                pt.sync()
            for r0 in range(0, nf, slab):
                k = min(slab, nf - r0)
                _ffi.check(lib.sdrk_memcpy_d2h(0, a.ctypes.data_as(ctypes.c_void_p), ctypes.c_void_p(d_a.value + r0 * N * 4), k * N * 4))
                _ffi.check(lib.sdrk_memcpy_d2h(0, b.ctypes.data_as(ctypes.c_void_p), ctypes.c_void_p(d_b.value + r0 * N * 4), k * N * 4))
                if not np.array_equal(a[:k].view(np.uint32), b[:k].view(np.uint32)):
                    bad = np.argwhere(a[:k].view(np.uint32) != b[:k].view(np.uint32))
                    raise AssertionError(f"call {c}: frames {nf} hop {hop} window kind {kind} shift {shift} eps {eps} offset {off}: "
                                         f"{len(bad)} values differ, first at row {r0 + bad[0][0]} bin {bad[0][1]}")
            done += 1
            if verbose:
                print(f"call {c}: {nf} frames, hop {hop}, window kind {kind}, shift {shift}, eps {eps}, x{reps}: identical", flush=True)
    finally:
        for d in (d_in, d_a, d_b):
            lib.sdrk_dev_free(0, d)
    return done


if __name__ == "__main__":
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 40
    s = int(sys.argv[2]) if len(sys.argv) > 2 else 1
    print("all identical:", run(n, s, verbose=True), "calls")
