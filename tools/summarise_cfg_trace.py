#!/usr/bin/env python3
"""Set a rocprofv3 kernel trace of tools/cfg_steady.py beside the event times the same process printed.

    python3 tools/summarise_cfg_trace.py TRACE_DIR STEADY_JSON [--transforms K]

The last K transforms of the trace (K * launches-per-transform dispatches of the col / row kernels, in start order) are
cut into transforms; per transform: sum of the col-pass durations, of the row-pass durations, their total, the span from
the first kernel's start to the last kernel's end, and the idle time between kernels inside the span.  Printed as one
JSON object with per-transform lists and medians, next to the HIP-event times of the same K transforms."""
import csv
import glob
import json
import os
import re
import statistics
import sys

trace_dir, steady = sys.argv[1], sys.argv[2]
K = int(sys.argv[sys.argv.index("--transforms") + 1]) if "--transforms" in sys.argv else None
rec = json.loads(open(steady).read().strip().splitlines()[-1])
K = K or len(rec["each_ms"])
files = glob.glob(os.path.join(trace_dir, "**", "*kernel_trace.csv"), recursive=True)
f = max(files, key=os.path.getmtime)
rows = []
for r in csv.DictReader(open(f)):
    m = re.search(r"sdrk::(\w+)(<[^>]*>)?", r["Kernel_Name"])
    name = (m.group(1) + (m.group(2) or "")) if m else r["Kernel_Name"][:40]
    if "col_pass" in name or "row_pass" in name or "fused64k" in name:     # (N = 65536, default plan: ONE persistent launch)
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), name))
rows.sort()
# launches per transform: from the run of the first kernel name repeating
total = len(rows)
n_warm = len(rec.get("warmup_ms", []))
per = total // (K + n_warm) if (K + n_warm) and total % (K + n_warm) == 0 else None
if per is None:
    raise SystemExit(f"{total} col/row dispatches do not divide into {K} + {n_warm} transforms")
tail = rows[-K * per:]
out = {"trace": os.path.relpath(f), "launches_per_transform": per, "transforms": K, "kernels": sorted({r[2] for r in tail}),
       "col_us": [], "row_us": [], "kernel_sum_ms": [], "span_ms": [], "idle_inside_span_us": []}
for t in range(K):
    ks = tail[t * per:(t + 1) * per]
    col = sum(e - s for s, e, n in ks if "col_pass" in n) / 1e3
    row = sum(e - s for s, e, n in ks if "row_pass" in n) / 1e3
    span = (ks[-1][1] - ks[0][0]) / 1e3
    out["col_us"].append(round(col, 1))
    out["row_us"].append(round(row, 1))
    out["kernel_sum_ms"].append(round((col + row) / 1e3, 4))
    out["span_ms"].append(round(span / 1e3, 4))
    fused = sum(e - s for s, e, n in ks if "fused64k" in n) / 1e3
    if fused:
        out.setdefault("fused_us", []).append(round(fused, 1))
        out["kernel_sum_ms"][-1] = round((col + row + fused) / 1e3, 4)
    out["idle_inside_span_us"].append(round(span - col - row - fused, 1))
chunks = max(per // 2, 1)
out["median"] = {"col_us_per_launch": round(statistics.median(out["col_us"]) / chunks, 2),
                 "row_us_per_launch": round(statistics.median(out["row_us"]) / chunks, 2),
                 "kernel_sum_ms": statistics.median(out["kernel_sum_ms"]), "span_ms": statistics.median(out["span_ms"])}
out["event_timed_ms_same_process"] = rec["each_ms"]
out["event_timed_median_ms"] = rec["median_ms"]
out["kernel_sum_over_event_time"] = round(out["median"]["kernel_sum_ms"] / rec["median_ms"], 4)
out["note"] = ("col/row per launch = per-transform sum / %d launches (the last, shorter chunk included); "
               "kernel_sum = sum of End - Start over the transform's dispatches" % chunks)
print(json.dumps(out))
