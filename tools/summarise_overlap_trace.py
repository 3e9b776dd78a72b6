#!/usr/bin/env python3
"""Digest of rocprofv3 --kernel-trace CSVs of tools/overlap_trace.py: per pass kernel the mean duration, and how
much of the row passes' time a col pass was running beside them.
    python3 tools/summarise_overlap_trace.py <dir with *_kernel_trace.csv> <label>"""
import csv, glob, sys
d, label = sys.argv[1], sys.argv[2]
rows = []
for f in glob.glob(d + "/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        name = r["Kernel_Name"]
        if "_pass_" not in name:
            continue
        kind = "col" if "col_pass" in name else "row"
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), kind))
rows.sort()
if not rows:
    sys.exit("no pass kernels in " + d)
for kind in ("col", "row"):
    dur = [(e - s) / 1e3 for s, e, k in rows if k == kind]
    print(f"{label}: {kind} pass  n={len(dur)}  mean {sum(dur)/len(dur):.1f} us  min {min(dur):.1f}  max {max(dur):.1f}")
cols = [(s, e) for s, e, k in rows if k == "col"]
tot = ov = 0
for s, e, k in rows:
    if k != "row":
        continue
    tot += e - s
    for cs, ce in cols:
        lo, hi = max(s, cs), min(e, ce)
        if hi > lo:
            ov += hi - lo
span = (max(e for s, e, k in rows) - min(s for s, e, k in rows)) / 1e3
busy = sum(e - s for s, e, k in rows) / 1e3
print(f"{label}: a col pass runs beside {100.0 * ov / tot:.1f} % of the row passes' time; "
      f"sum of kernel durations {busy/1e3:.2f} ms over a span of {span/1e3:.2f} ms (all launches of the run)")
