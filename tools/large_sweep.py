#!/usr/bin/env python3
"""Developer sweep: large-frame configs vs scratch chunk size (SDRK_SCRATCH_MB)."""
import json, os, subprocess, sys
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for mb in sys.argv[1:] or ["16", "32", "64", "128", "256", "1024"]:
    env = dict(os.environ, SDRK_SCRATCH_MB=mb)
    out = subprocess.run([sys.executable, os.path.join(REPO, "tools", "extra_bench.py"), "--large-only"],
                         env=env, capture_output=True, text=True).stdout
    for line in out.splitlines():
        d = json.loads(line)
        if "nfft" in d and d["nfft"] > 4096:
            print(mb, "MiB:", d["what"][:40], d["ms"], "ms", d["algorithmic_GBps"], "GB/s", flush=True)
