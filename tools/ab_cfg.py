#!/usr/bin/env python3
"""Developer helper (GPU box): A/B of library variants (tools/variant.sh) on the large-frame configurations.
Every (variant, configuration) runs in its own process `rounds` times, interleaved, so that the allocation-placement
levels average out; prints min / median of the per-process medians.

    python3 tools/ab_cfg.py [--rounds 4] [--cfg 3|5|both] [--env SDRK_SCRATCH_MB=96] base st16 ld2 ...
"""
import argparse, json, os, statistics, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ap = argparse.ArgumentParser()
ap.add_argument("--rounds", type=int, default=4)
ap.add_argument("--cfg", default="both")
ap.add_argument("--env", action="append", default=[])
ap.add_argument("variants", nargs="+")
a = ap.parse_args()
cfgs = {"3": ["65536", "18749", "32768", "hann"], "5": ["1048576", "256", "1048576", "hann"]}
use = ["3", "5"] if a.cfg == "both" else [a.cfg]
res = {}
for r in range(a.rounds):
    for v in a.variants:
        lib = os.path.join(ROOT, "sdr-iq-visualizer_amd", "lib" if v == "base" else "lib_" + v, "libsdrk.so")
        env = dict(os.environ, SDRK_LIB=lib)
        for kv in a.env:
            k, val = kv.split("=", 1)
            env[k] = val
        for c in use:
            out = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "one_config.py")] + cfgs[c], env=env,
                                 capture_output=True, text=True, timeout=300)
            if out.returncode != 0:
                print(f"{v} cfg{c}: FAILED rc={out.returncode}: {out.stderr[-300:]}", flush=True)
                sys.exit(1)
            res.setdefault((v, c), []).append(json.loads(out.stdout.strip().splitlines()[-1])["ms"])
print("env:", " ".join(a.env) or "-")
for c in use:
    for v in a.variants:
        ms = res[(v, c)]
        print(f"cfg{c} {v:12s} min {min(ms):.3f}  median {statistics.median(ms):.3f}  all {' '.join('%.3f' % m for m in ms)}", flush=True)
