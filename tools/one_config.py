#!/usr/bin/env python3
"""Developer helper: run one device-resident large-frame configuration a few times (for rocprofv3)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tools.extra_bench import device_run
import json
nfft = int(sys.argv[1]); frames = int(sys.argv[2]); hop = int(sys.argv[3]); win = sys.argv[4] if len(sys.argv) > 4 else "hann"
print(json.dumps(device_run(nfft, frames, hop, None if win == "rect" else win, 3, "one_config")))
