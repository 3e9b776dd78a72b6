#!/usr/bin/env python3
"""the reference's numpy path on the host cores of the GPU box — per-frame loop (1 process),
batched oracle (1 process), and the batched oracle under multiprocessing with one worker per core.
Pure CPU (no GPU runtime is touched, so forking workers is safe).  Prints one JSON object."""
import json, multiprocessing as mp, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np

NFFT, CHUNK = 4096, 128


def _worker(args):
    seed, n_chunks, seconds = args
    from oracle import cpu_ref
    from sdr_iq_visualizer_amd import synth
    frames = [synth.synth_iq(seed, c * CHUNK, CHUNK, NFFT) for c in range(n_chunks)]
    w = np.hanning(NFFT).astype(np.float32)
    cpu_ref.spectrum_db(frames[0], window=w)
    done, t0 = 0, time.perf_counter()
    while time.perf_counter() - t0 < seconds:
        for f in frames:
            cpu_ref.spectrum_db(f, window=w)
        done += CHUNK * n_chunks
    return done, time.perf_counter() - t0


def main():
    seconds = float(sys.argv[1]) if len(sys.argv) > 1 else 5.0
    cores = os.cpu_count() or 1
    try:
        cores = len(os.sched_getaffinity(0))
    except AttributeError:
        pass
    done, dt = _worker((1234, 8, seconds))
    single = done * NFFT / dt / 1e6
    from sdr_iq_visualizer_amd import synth
    x = synth.synth_iq(1234, 0, CHUNK, NFFT)
    t0, k = time.perf_counter(), 0
    while time.perf_counter() - t0 < min(seconds, 3.0):
        for f in x:
            20 * np.log10(np.abs(np.fft.fftshift(np.fft.fft(f))) + 1e-12)
        k += CHUNK
    per_frame = k * NFFT / (time.perf_counter() - t0) / 1e6
    with mp.get_context("fork").Pool(cores) as pool:
        t0 = time.perf_counter()
        res = pool.map(_worker, [(1234 + i, 4, seconds) for i in range(cores)])
        wall = time.perf_counter() - t0
    total = sum(r[0] for r in res) * NFFT / max(r[1] for r in res) / 1e6
    print(json.dumps({"what": "numpy oracle on host cores, N=4096 Hann (BASELINE.md §3)", "numpy": np.__version__,
                      "cores_used": cores, "cpu_count": os.cpu_count(),
                      "per_frame_reference_expression_rect_Msps_1core": round(per_frame, 1),
                      "batched_oracle_Msps_1core": round(single, 1),
                      "batched_oracle_Msps_all_cores": round(total, 1), "wall_s": round(wall, 1)}))


if __name__ == "__main__":
    main()
