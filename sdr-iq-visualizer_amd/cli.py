"""Offline PSD of a SigMF recording on the GPU (BASELINE.json config 1).

    python -m sdr_iq_visualizer_amd.cli psd recording.sigmf-meta [--nfft 4096] [--welch 1024] [--out rows.npz]
    python -m sdr_iq_visualizer_amd.cli synth out_base --frames 8 --nfft 4096      # write a test recording

``psd`` reproduces, for the first ``--nfft`` samples, the reference's live expression
(app/sdr/streamer.py:119-121) and, with ``--welch N``, the averaged Hann PSD its offline script
plots (scripts/process_sigmf_data.py:188-189).  All transforms run through libsdrk.
"""
from __future__ import annotations

import argparse
import json
import sys

import numpy as np


def main(argv=None) -> int:
    ap = argparse.ArgumentParser(prog="sdr_iq_visualizer_amd.cli")
    sub = ap.add_subparsers(dest="cmd", required=True)
    p = sub.add_parser("psd")
    p.add_argument("path")
    p.add_argument("--nfft", type=int, default=4096)
    p.add_argument("--welch", type=int, default=0, help="also compute the averaged PSD with this NFFT (Hann)")
    p.add_argument("--window", default=None)
    p.add_argument("--device", type=int, default=0)
    p.add_argument("--out", default=None, help="write results to this .npz")
    s = sub.add_parser("synth")
    s.add_argument("base")
    s.add_argument("--frames", type=int, default=8)
    s.add_argument("--nfft", type=int, default=4096)
    s.add_argument("--sample-rate", type=float, default=1_000_000)
    s.add_argument("--center-freq", type=float, default=2_400_000_000)
    s.add_argument("--seed", type=int, default=1234)
    args = ap.parse_args(argv)

    from . import sigmf_io, synth
    if args.cmd == "synth":
        x = synth.synth_iq(args.seed, 0, args.frames, args.nfft).reshape(-1)
        paths = sigmf_io.write_sigmf(args.base, x, args.sample_rate, args.center_freq,
                                     description="synthetic 12-bit IQ (sdr_iq_visualizer_amd.synth)")
        print(json.dumps({"wrote": paths, "samples": int(x.size)}))
        return 0

    from . import spectrum
    samples, meta = sigmf_io.read_sigmf(args.path)
    if samples.size < args.nfft:
        print(f"recording has {samples.size} samples, need {args.nfft}", file=sys.stderr)
        return 2
    fs, fc = meta["sample_rate"], meta["center_freq"]
    power_db = spectrum.spectrum_db(samples[: args.nfft], window=args.window, device=args.device)
    freqs = spectrum.freq_axis(args.nfft, fs, fc)
    k = int(np.argmax(power_db))
    report = {"samples": int(samples.size), "sample_rate": fs, "center_freq": fc, "nfft": args.nfft,
              "peak_db": float(power_db[k]), "peak_freq_hz": float(freqs[k]),
              "median_db": float(np.median(power_db))}
    results = {"power_db": power_db, "freqs": freqs}
    if args.welch:
        pxx = spectrum.welch_psd(samples, args.welch, fs, device=args.device)
        results["welch_pxx"] = pxx
        results["welch_freqs"] = spectrum.freq_axis(args.welch, fs, fc)
        report["welch_nfft"] = args.welch
        report["welch_segments"] = 1 + (samples.size - args.welch) // args.welch
        report["welch_peak_db_per_hz"] = float(10 * np.log10(pxx.max()))
    if args.out:
        np.savez_compressed(args.out, **results)
        report["out"] = args.out
    print(json.dumps(report))
    return 0


if __name__ == "__main__":
    sys.exit(main())
