#!/usr/bin/env python3
"""Kernel trace of tools/channel_probe.py -> per run of the channel (256 frames): kernel time by kind, and the GPU's idle time
between kernels (host-bound gaps), for the LAST `runs` pipelined runs.

    python3 tools/summarise_channel_trace.py TRACE_DIR [--batches 11] [--runs 5]"""
import csv, glob, json, os, re, statistics, sys
d = sys.argv[1]
batches = int(sys.argv[sys.argv.index("--batches") + 1]) if "--batches" in sys.argv else 11
runs = int(sys.argv[sys.argv.index("--runs") + 1]) if "--runs" in sys.argv else 5
f = max(glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True), key=os.path.getmtime)
rows = []
for r in csv.DictReader(open(f)):
    m = re.search(r"sdrk::(\w+)", r["Kernel_Name"])
    if m and m.group(1) != "synth_fill_kernel":
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), m.group(1)))
rows.sort()
# a run = `batches` decimate launches; cut at the decimate kernels, from the end
dec = [i for i, r in enumerate(rows) if r[2].startswith("decimate")]
out = {"trace": os.path.relpath(f), "runs": []}
for k in range(runs):
    hi = dec[len(dec) - 1 - k * batches]
    lo = dec[len(dec) - 1 - (k + 1) * batches] + 1
    seg = rows[lo:hi + 1]
    by = {}
    for s, e, n in seg:
        by[n] = by.get(n, 0) + (e - s) / 1e3
    span = (seg[-1][1] - seg[0][0]) / 1e3
    busy = sum(by.values())
    out["runs"].append({"launches": len(seg), "span_us": round(span, 1), "kernel_us": {n: round(v, 1) for n, v in by.items()},
                        "kernel_sum_us": round(busy, 1), "gpu_idle_inside_span_us": round(span - busy, 1)})
out["median_span_us"] = statistics.median(r["span_us"] for r in out["runs"])
out["median_kernel_sum_us"] = statistics.median(r["kernel_sum_us"] for r in out["runs"])
out["median_idle_us"] = statistics.median(r["gpu_idle_inside_span_us"] for r in out["runs"])
print(json.dumps(out, indent=1))
