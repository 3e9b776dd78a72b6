"""Developer helper (GPU box): randomised large-frame cases (length, frame count, hop, window, shift, eps, epilogue)
against the oracle on sampled frames.  python tools/stress_large.py [cases] [seed]"""
import ctypes, sys
import numpy as np
sys.path.insert(0, __file__.rsplit("/", 2)[0])
from oracle import cpu_ref
from sdr_iq_visualizer_amd import _ffi, synth
from sdr_iq_visualizer_amd.spectrum import SpectrumPlan
from tests.parity import assert_complex_parity, assert_db_parity

cases = int(sys.argv[1]) if len(sys.argv) > 1 else 40
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
lib = _ffi.lib()
for c in range(cases):
    lg = int(rng.integers(15, 23))
    n = 1 << lg
    rows = int(rng.integers(1, max(2, min(60, (1 << 27) // n))))
    hop = n if rng.random() < 0.4 else int(rng.integers(n // 4, n + 1))
    window = None if rng.random() < 0.5 else "hann"
    shift = bool(rng.random() < 0.8)
    eps = float(rng.choice([1e-12, 1e-10, 0.0]))
    L = n + (rows - 1) * hop
    gen = (L + 4095) // 4096
    d_in, d_out = ctypes.c_void_p(), ctypes.c_void_p()
    _ffi.check(lib.sdrk_dev_alloc(0, gen * 4096 * 8, ctypes.byref(d_in)))
    _ffi.check(lib.sdrk_dev_alloc(0, rows * n * 4, ctypes.byref(d_out)))
    try:
        _ffi.check(lib.sdrk_synth_fill(0, 100 + c, 0, gen, 4096, d_in, None))
        with SpectrumPlan(n, window=window, shift=shift, eps=eps) as plan:
            plan.exec_device(d_in.value, rows, d_out.value, frame_stride=hop)
            plan.sync()
            stream = synth.synth_iq(100 + c, 0, gen, 4096).reshape(-1)
            picks = sorted({0, rows - 1, rows // 2, int(rng.integers(0, rows))})
            got = np.empty((len(picks), n), np.float32)
            for i, r in enumerate(picks):
                _ffi.check(lib.sdrk_memcpy_d2h(0, got[i].ctypes.data_as(ctypes.c_void_p), ctypes.c_void_p(d_out.value + r * n * 4), n * 4))
            frames = np.stack([stream[r * hop: r * hop + n] for r in picks])
            w = np.hanning(n) if window else None
            assert_db_parity(got, cpu_ref.spectrum_db(frames, window=w, eps=eps, shift=shift), what=f"case {c}")
            if rng.random() < 0.3:          # the complex epilogue through the host path
                x = frames[:2]
                assert_complex_parity(plan.fft(x), cpu_ref.fft(x, window=w, shift=shift), what=f"case {c} fft")
        print(f"case {c:3d} ok: N=2^{lg} rows={rows} hop={hop} window={window} shift={shift} eps={eps}", flush=True)
    finally:
        lib.sdrk_dev_free(0, d_in)
        lib.sdrk_dev_free(0, d_out)
print("all ok")
