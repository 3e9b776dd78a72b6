"""Developer helper (GPU box): randomised calls through the numpy boundary (sdrk_exec_host / sdrk_exec_fft_host): frame
length, frame count, hop (gaps and overlaps), window, shift, eps, pageable or pinned arrays on either side, reused
result arrays, the complex epilogue — against the oracle on sampled frames.  python tools/stress_host.py [cases] [seed]"""
import sys
import numpy as np
sys.path.insert(0, __file__.rsplit("/", 2)[0])
from oracle import cpu_ref
import sdr_iq_visualizer_amd as pkg
from sdr_iq_visualizer_amd.spectrum import SpectrumPlan
from tests.parity import assert_complex_parity, assert_db_parity


def run(cases, seed):
    rng = np.random.default_rng(seed)
    for c in range(cases):
        lg = int(rng.integers(1, 18))
        n = 1 << lg if rng.random() < 0.85 else int(rng.integers(3, 5000))
        budget = 1 << int(rng.integers(12, 25))                       # samples per call: 32 KiB ... 128 MiB of input
        hop = n if rng.random() < 0.5 else int(rng.integers(1, 2 * n + 1))
        rows = int(max(1, min(6000, budget // max(hop, 1))))
        L = n + (rows - 1) * hop
        if L > (1 << 25):
            rows = max(1, ((1 << 25) - n) // hop + 1)
            L = n + (rows - 1) * hop
        window = None if rng.random() < 0.5 else "hann"
        shift = bool(rng.random() < 0.8)
        eps = float(rng.choice([1e-12, 1e-10, 0.0]))
        pin_in, pin_out = bool(rng.random() < 0.4), bool(rng.random() < 0.4)
        x = ((rng.standard_normal(L) + 1j * rng.standard_normal(L)) * rng.uniform(0.01, 2000)).astype(np.complex64)
        if pin_in:
            xp = pkg.pinned_empty(x.shape, np.complex64)
            xp[...] = x
        else:
            xp = x
        w = np.hanning(n) if window else None
        with SpectrumPlan(n, window=window, shift=shift, eps=eps) as plan:
            out = pkg.pinned_empty((rows, n), np.float32) if pin_out else np.full((rows, n), np.nan, np.float32)
            if hop == n:
                got = plan.spectrum_db(xp.reshape(rows, n), out=out)
            else:
                got = plan.stft_db(xp, hop, out=out)
            assert got is out and got.shape == (rows, n)
            picks = sorted({0, rows - 1, rows // 2, int(rng.integers(0, rows)), int(rng.integers(0, rows))})
            frames = np.stack([x[r * hop: r * hop + n] for r in picks])
            assert_db_parity(got[picks], cpu_ref.spectrum_db(frames, window=w, eps=eps, shift=shift), what=f"case {c}")
            assert np.all(np.isfinite(got) | (eps == 0.0)), f"case {c}: unwritten rows"
            if hop == n and rng.random() < 0.4:
                k = min(rows, 3)
                assert_complex_parity(plan.fft(xp.reshape(rows, n)[:k]), cpu_ref.fft(x.reshape(rows, n)[:k], window=w, shift=shift),
                                      what=f"case {c} fft")
        print(f"case {c:3d} ok: N={n} rows={rows} hop={hop} window={window} shift={shift} eps={eps} pinned in/out={pin_in}/{pin_out}", flush=True)
    return cases


if __name__ == "__main__":
    run(int(sys.argv[1]) if len(sys.argv) > 1 else 100, int(sys.argv[2]) if len(sys.argv) > 2 else 1)
    print("all ok")
