#!/usr/bin/env python3
"""Copy what tools/collect_round.sh left under gpurun_out/ into profiles/<tag>/ (tracked) and regenerate
profiles/hbm_traffic.json from the fresh PMC summary.
    python tools/install_profiles.py r02"""
import json, os, shutil, subprocess, sys

tag = sys.argv[1] if len(sys.argv) > 1 else "r02"
repo = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src, prof, dst = (os.path.join(repo, "gpurun_out", f"collect_{tag}"), os.path.join(repo, "gpurun_out", f"prof_{tag}"),
                  os.path.join(repo, "profiles", tag))
os.makedirs(dst, exist_ok=True)
for name in ("bench_n1.json", "bench_kernel_stats.csv", "cfg3_kernel_stats.csv", "cfg5_kernel_stats.csv", "size_sweep.log",
             "placeprobe.log", "placement_16_runs.txt", "plain_alloc_8_runs.txt", "fused64k_vs_tiled.log"):
    if os.path.exists(os.path.join(src, name)):
        shutil.copy(os.path.join(src, name), os.path.join(dst, name))
# summary and rows from the newest rocprofv3 files (gpurun merges every call into the same local directories)
summ = subprocess.run([sys.executable, os.path.join(repo, "tools", "summarise_profiles.py"), prof], check=True,
                      capture_output=True, text=True).stdout
open(os.path.join(dst, "rocprof_summary.json"), "w").write(summ)
subprocess.run([sys.executable, os.path.join(repo, "tools", "summarise_profiles.py"), prof, "--rows",
                os.path.join(dst, "pmc_fetch_write_rows.csv")], check=True)
k = next(v | {"name": n} for n, v in json.loads(summ)["bench"].items() if n.startswith("fft4096_kernel"))
frames, nfft = 1 << 20, 4096
traffic = {
    "frames": frames, "window": "hann", "nfft": nfft, "kernel": "sdrk::" + k["name"],
    "FETCH_SIZE_KB_mean": k["FETCH_SIZE_KB_mean"], "WRITE_SIZE_KB_mean": k["WRITE_SIZE_KB_mean"],
    "read_bytes_corrected": k["read_bytes_per_dispatch"], "write_bytes": k["write_bytes_per_dispatch"],
    "bytes_per_launch": k["read_bytes_per_dispatch"] + k["write_bytes_per_dispatch"],
    "algorithmic_bytes_per_launch": frames * nfft * 12,
    "correction": "gfx950: FETCH_SIZE counts 128-B requests as 64 B (MI355X_MICROARCH.md, HBM section) -> read bytes = "
                  "2 x FETCH_SIZE x 1024; WRITE_SIZE x 1024 is exact",
    "source": f"profiles/{tag}/pmc_fetch_write_rows.csv (separate --pmc FETCH_SIZE / --pmc WRITE_SIZE passes of python3 bench.py "
              f"--steps 3 --warmup 1 --cpu-seconds 0 --parity-frames 0 --no-secondary; tools/profile_round.sh {tag})",
}
json.dump(traffic, open(os.path.join(repo, "profiles", "hbm_traffic.json"), "w"), indent=1)
print(json.dumps(traffic, indent=1))
