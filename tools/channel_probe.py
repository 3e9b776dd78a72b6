#!/usr/bin/env python3
"""Developer helper (GPU box): BASELINE.json configs[4] as worded — one continuous N = 2^20 channel with ring and decimated
host gather (bench.py's channel_config5) — alone in a process, for a rocprofv3 kernel trace of that leg.

    python3 tools/channel_probe.py [--batch 24] [--frames 256]"""
import argparse
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
ap = argparse.ArgumentParser()
ap.add_argument("--batch", type=int, default=24)
ap.add_argument("--frames", type=int, default=256)
a = ap.parse_args()
import bench  # noqa: E402
import sdr_iq_visualizer_amd as pkg  # noqa: E402
from sdr_iq_visualizer_amd import _ffi  # noqa: E402
from sdr_iq_visualizer_amd.spectrum import SpectrumPlan  # noqa: E402

rec = bench.channel_config5(_ffi.lib(), _ffi, pkg, SpectrumPlan, 0, n_frames=a.frames, batch=a.batch)
print(json.dumps(rec))
