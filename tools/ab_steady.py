#!/usr/bin/env python3
"""Developer helper (GPU box): A/B of library variants (tools/variant.sh) on one large-frame configuration in the WARM
state — every run is tools/cfg_steady.py in its own process (100 ms of the plan's own launches first, then K transforms
back to back), variants interleaved, so neither the clock ramp of a cold start nor the allocation-placement levels
 decide the comparison.  (tools/ab_cfg.py's one warm-up + three isolated launches measure the first
few milliseconds after idle, at 1.0-1.4 GHz instead of 1.85-2.0: round 5 found its numbers 15 % above the steady state.)

    python3 tools/ab_steady.py [--rounds 3] [--cfg "1048576 256 1048576 hann"] [--env K=V] base e1 ..."""
import argparse, json, os, statistics, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ap = argparse.ArgumentParser()
ap.add_argument("--rounds", type=int, default=3)
ap.add_argument("--cfg", default="1048576 256 1048576 hann")
ap.add_argument("--transforms", type=int, default=20)
ap.add_argument("--env", action="append", default=[])
ap.add_argument("variants", nargs="+")
a = ap.parse_args()
res = {}
for r in range(a.rounds):
    for v in a.variants:
        name, _, envs = v.partition(":")                    # "base:SDRK_SCRATCH_MB=128" = variant with its own environment
        lib = os.path.join(ROOT, "sdr-iq-visualizer_amd", "lib" if name == "base" else "lib_" + name, "libsdrk.so")
        env = dict(os.environ, SDRK_LIB=lib)
        for kv in a.env + ([envs] if envs else []):
            k, val = kv.split("=", 1)
            env[k] = val
        out = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "cfg_steady.py")] + a.cfg.split() +
                             ["--transforms", str(a.transforms)], env=env, capture_output=True, text=True, timeout=300)
        if out.returncode != 0:
            print(f"{v}: FAILED rc={out.returncode}: {out.stderr[-300:]}", flush=True)
            sys.exit(1)
        d = json.loads(out.stdout.strip().splitlines()[-1])
        res.setdefault(v, []).append((d["median_ms"], d["min_ms"], d["telemetry"]["sclk_mhz"]["mean"] if d["telemetry"]["sclk_mhz"] else None))
print("cfg:", a.cfg, " env:", " ".join(a.env) or "-")
for v in a.variants:
    ms = [m for m, _, _ in res[v]]
    print(f"{v:28s} median-of-medians {statistics.median(ms):.4f}  min {min(m for _, m, _ in res[v]):.4f}  runs "
          + " ".join("%.4f" % m for m in ms) + "  sclk " + " ".join(str(s) for _, _, s in res[v]), flush=True)
