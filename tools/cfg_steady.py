#!/usr/bin/env python3
"""Developer helper (GPU box): one device-resident large-frame configuration measured the way bench.py measures it — warmed
up BY TIME, then K transforms back to back with a HIP event between consecutive ones — with the clock / power sampler of
bench.py running beside it.  Run once plain and once under `rocprofv3 --kernel-trace` (tools/summarise_cfg_trace.py then
sets the per-transform sums of the kernel trace beside the event times this process printed), so that a profile figure
can be held against the un-profiled step it is supposed to explain.

    python3 tools/cfg_steady.py NFFT FRAMES HOP [hann|rect] [--warm-ms 100] [--transforms 20] [--isolated] [--out FILE]

--isolated: a host synchronisation after every transform (what tools/one_config.py and the placement probes do).
Prints one JSON object: each_ms (the K transforms in order), warm-up series, telemetry summary and the raw samples
(time since start in ms, sclk, mclk, fclk, socket power) of the timed region."""
import argparse
import ctypes
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

ap = argparse.ArgumentParser()
ap.add_argument("nfft", type=int)
ap.add_argument("frames", type=int)
ap.add_argument("hop", type=int)
ap.add_argument("window", nargs="?", default="hann")
ap.add_argument("--warm-ms", type=float, default=100.0)
ap.add_argument("--transforms", type=int, default=20)
ap.add_argument("--isolated", action="store_true")
ap.add_argument("--idle-ms", type=float, default=0.0, help="sleep this long between the warm-up and the timed transforms")
ap.add_argument("--out", default=None)
ap.add_argument("--form", choices=["default", "tiled", "fused"], default="default",
                help="nfft = 65536 only: the plan's default (the persistent launch from 512 frames), the two tiled launches, or "
                     "the persistent launch for every call")
a = ap.parse_args()

from bench import Telemetry  # noqa: E402  (the sampler only; nothing of the bench runs)
from sdr_iq_visualizer_amd import _ffi  # noqa: E402
from sdr_iq_visualizer_amd.spectrum import SpectrumPlan  # noqa: E402

lib = _ffi.lib()
dev = 0
window = None if a.window == "rect" else a.window
in_samples = (a.frames - 1) * a.hop + a.nfft
gen_frames = (in_samples + 4095) // 4096
d_gen, d_out = ctypes.c_void_p(), ctypes.c_void_p()
_ffi.check(lib.sdrk_dev_alloc(dev, gen_frames * 4096 * 8, ctypes.byref(d_gen)))
_ffi.check(lib.sdrk_dev_alloc(dev, a.frames * a.nfft * 4, ctypes.byref(d_out)))
_ffi.check(lib.sdrk_synth_fill(dev, 99, 0, gen_frames, 4096, d_gen, None))
tel = Telemetry(_ffi.device_info(dev))
rec = {"nfft": a.nfft, "frames": a.frames, "hop": a.hop, "window": a.window, "isolated": a.isolated,
       "library": _ffi.library_path(), "profiled": bool(os.environ.get("ROCPROFILER_LIBRARY_CTOR") or os.environ.get("ROCP_TOOL_LIBRARIES")
                                                      or "rocprof" in os.environ.get("LD_PRELOAD", ""))}
rec["form"] = a.form
with SpectrumPlan(a.nfft, window=window, device=dev, fused64k={"default": None, "tiled": False, "fused": True}[a.form]) as plan:
    warm = []
    t0 = time.perf_counter()
    while (time.perf_counter() - t0) * 1e3 < a.warm_ms or len(warm) < 3:
        warm.append(plan.exec_device_timed(d_gen.value, a.frames, d_out.value, 1, frame_stride=a.hop))
    rec["warmup_ms"] = [round(float(v), 4) for v in warm]
    if a.idle_ms > 0:
        time.sleep(a.idle_ms * 1e-3)
    tel.start(period=0.0005)
    t_start = time.perf_counter()
    if a.isolated:
        each = [plan.exec_device_timed(d_gen.value, a.frames, d_out.value, 1, frame_stride=a.hop) for _ in range(a.transforms)]
    else:
        each = plan.exec_device_timed_each(d_gen.value, a.frames, d_out.value, a.transforms, frame_stride=a.hop)
    wall = time.perf_counter() - t_start
    tel.stop()
rec["each_ms"] = [round(float(v), 4) for v in each]
s = sorted(rec["each_ms"])
rec["median_ms"], rec["min_ms"], rec["max_ms"] = s[len(s) // 2], s[0], s[-1]
rec["wall_ms_per_transform"] = round(wall * 1e3 / a.transforms, 4)
rec["telemetry"] = tel.summary()
rec["telemetry_samples"] = [[s.get("sclk_mhz"), s.get("mclk_mhz"), s.get("fclk_mhz"), s.get("power_w")] for s in tel.samples][:400]
lib.sdrk_dev_free(dev, d_gen)
lib.sdrk_dev_free(dev, d_out)
txt = json.dumps(rec)
if a.out:
    open(a.out, "w").write(txt + "\n")
brief = {k: v for k, v in rec.items() if k != "telemetry_samples"}
print(json.dumps(brief))
