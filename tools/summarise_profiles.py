#!/usr/bin/env python3
"""Turn the rocprofv3 output directories written by tools/profile_round.sh into one JSON summary:
per kernel the call count and average duration (kernel-trace --stats) and the mean FETCH_SIZE / WRITE_SIZE
per dispatch with the gfx950 correction applied (read bytes = 2 x FETCH_SIZE KB x 1024: the counter tallies
each 128-byte request as 64 bytes; WRITE_SIZE KB x 1024 is exact — MI355X_MICROARCH.md, HBM section).
Infinity-Cache hits are included in both counters (they count L2 <-> fabric requests)."""
import csv, glob, json, os, re, sys
from collections import defaultdict


def short(name):
    m = re.search(r"sdrk::(\w+)(<[^>]*>)?", name)
    return (m.group(1) + (m.group(2) or "")) if m else name[:60]


def newest(dirname, pattern):
    """The most recent file of one rocprofv3 output directory (gpurun merges every call's files into the same local
    directories; on the GPU box there is only one)."""
    files = glob.glob(os.path.join(dirname, "**", pattern), recursive=True)
    return [max(files, key=os.path.getmtime)] if files else []


def stats(dirname):
    out = {}
    for f in newest(dirname, "*kernel_stats.csv"):
        for row in csv.DictReader(open(f)):
            out[short(row["Name"])] = {"calls": int(row["Calls"]), "avg_us": round(float(row["AverageNs"]) / 1e3, 3),
                                       "min_us": round(float(row["MinNs"]) / 1e3, 3), "max_us": round(float(row["MaxNs"]) / 1e3, 3),
                                       "total_ms": round(float(row["TotalDurationNs"]) / 1e6, 4)}
    return out


def timed_launches(dirname, kernel_prefix, last_n):
    """From the kernel trace itself (not the --stats average): the LAST `last_n` dispatches of the kernel — bench.py's
    timed region; what comes before them are the placement probe's launches over candidate pairings and the warm-up."""
    for f in newest(dirname, "*kernel_trace.csv"):
        rows = [r for r in csv.DictReader(open(f)) if short(r["Kernel_Name"]).startswith(kernel_prefix)]
        rows.sort(key=lambda r: int(r["Start_Timestamp"]))
        if len(rows) < last_n:
            return None
        dur = [(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3 for r in rows]
        t, e = dur[-last_n:], dur[:-last_n]
        return {"timed_launches": last_n, "timed_avg_us": round(sum(t) / len(t), 3), "timed_min_us": round(min(t), 3),
                "timed_max_us": round(max(t), 3), "earlier_launches": len(e),
                "earlier_avg_us": round(sum(e) / len(e), 3) if e else None}
    return None


def pmc(dirname, counter):
    acc = defaultdict(list)
    for f in newest(dirname, "*counter_collection.csv"):
        for row in csv.DictReader(open(f)):
            if row["Counter_Name"] == counter:
                acc[short(row["Kernel_Name"])].append(float(row["Counter_Value"]))
    return {k: {"dispatches": len(v), "mean_KB": sum(v) / len(v), "sum_KB": sum(v)} for k, v in acc.items()}


def tags(root):
    """Every <tag>_trace directory of the collection, the BASELINE ones first."""
    found = sorted(os.path.basename(d)[:-6] for d in glob.glob(os.path.join(root, "*_trace")) if os.path.isdir(d))
    first = [t for t in ("bench", "cfg3", "cfg5") if t in found]
    return first + [t for t in found if t not in first]


def sq(root, tag):
    """SQ counters of the <tag>_sq* passes: per kernel the mean per dispatch."""
    acc = defaultdict(lambda: defaultdict(list))
    for d in sorted(glob.glob(os.path.join(root, f"{tag}_sq*"))):
        for f in newest(d, "*counter_collection.csv"):
            per = defaultdict(float)
            for row in csv.DictReader(open(f)):
                per[(row["Dispatch_Id"], short(row["Kernel_Name"]), row["Counter_Name"])] += float(row["Counter_Value"])
            for (_, k, c), v in per.items():
                acc[k][c].append(v)
    return {k: {c: round(sum(v) / len(v)) for c, v in sorted(cs.items())} | {"dispatches": max(len(v) for v in cs.values())}
            for k, cs in acc.items() if k.startswith(("fft4096", "row_"))}


def main(root):
    res = {}
    for tag in tags(root):
        st, fe, wr = stats(f"{root}/{tag}_trace"), pmc(f"{root}/{tag}_fetch", "FETCH_SIZE"), pmc(f"{root}/{tag}_write", "WRITE_SIZE")
        kernels = {}
        for k in sorted(set(st) | set(fe) | set(wr)):
            rec = dict(st.get(k, {}))
            if k in fe:
                rec["FETCH_SIZE_KB_mean"] = round(fe[k]["mean_KB"], 3)
                rec["read_bytes_per_dispatch"] = round(2 * fe[k]["mean_KB"] * 1024)
                rec["pmc_dispatches"] = fe[k]["dispatches"]
            if k in wr:
                rec["WRITE_SIZE_KB_mean"] = round(wr[k]["mean_KB"], 3)
                rec["write_bytes_per_dispatch"] = round(wr[k]["mean_KB"] * 1024)
            kernels[k] = rec
        if tag == "bench":       # python3 bench.py --no-secondary --cpu-seconds 0: 20 timed steps (the default --steps)
            for k in kernels:
                if k.startswith("fft4096_kernel"):
                    tl = timed_launches(f"{root}/{tag}_trace", "fft4096_kernel", 20)
                    if tl:
                        kernels[k].update(tl)
        counters = sq(root, tag)
        for k, c in counters.items():
            kernels.setdefault(k, {})["sq_counters_mean_per_dispatch"] = c
        res[tag] = kernels
    print(json.dumps(res, indent=1))


def rows(root, out_csv):
    """Every FETCH_SIZE / WRITE_SIZE row of the library's kernels from the six PMC passes, one CSV."""
    cols = ["run", "Dispatch_Id", "Kernel", "Grid_Size", "Workgroup_Size", "LDS_Block_Size", "VGPR_Count", "Counter_Name",
            "Counter_Value_KB", "Start_ns", "End_ns"]
    with open(out_csv, "w", newline="") as fh:
        w = csv.writer(fh)
        w.writerow(cols)
        for tag in tags(root):
            for kind in ("fetch", "write"):
                for f in newest(f"{root}/{tag}_{kind}", "*counter_collection.csv"):
                    for r in csv.DictReader(open(f)):
                        if "sdrk::" not in r["Kernel_Name"]:
                            continue
                        name = re.sub(r"\(.*", "", r["Kernel_Name"]).replace("void ", "")
                        w.writerow([tag, r["Dispatch_Id"], name, r["Grid_Size"], r["Workgroup_Size"], r["LDS_Block_Size"],
                                    r["VGPR_Count"], r["Counter_Name"], r["Counter_Value"], r["Start_Timestamp"], r["End_Timestamp"]])


if __name__ == "__main__":
    root = sys.argv[1] if len(sys.argv) > 1 else "gpurun_out/prof_r02"
    if len(sys.argv) > 3 and sys.argv[2] == "--rows":
        rows(root, sys.argv[3])
    else:
        main(root)
