#!/usr/bin/env python3
"""Device-resident throughput of every supported frame length (developer sweep)."""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tools.extra_bench import device_run
sizes = [int(a) for a in sys.argv[1:]] or [1 << k for k in range(4, 23)] + [1000, 5000, 100000]
for n in sizes:
    frames = max(1, (1 << 27) // n)
    r = device_run(n, frames, n, None, 5, "sweep")
    print(json.dumps({k: r[k] for k in ("nfft", "frames", "ms", "frame_Msamples_per_s", "algorithmic_GBps", "hbm_peak_frac")}), flush=True)
