"""Developer helper (GPU box): the staged (pageable) numpy boundary at B = 32768 and 65536 for several sizes of the copy
pool (SDRK_HOST_THREADS), each in its own process.  python3 tools/host_threads_probe.py [threads ...]"""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CHILD = r'''
import os, sys, time, numpy as np
if os.environ.get('PROBE_TORCH_FIRST'): import torch
sys.path.insert(0, %r)
import sdr_iq_visualizer_amd as pkg
from sdr_iq_visualizer_amd import synth
x = synth.synth_iq(1, 0, 32768, 4096)
for b in (32768, 65536):
    if b > x.shape[0]: x = np.concatenate([x, x])
    res = np.empty(x.shape, np.float32)
    pkg.spectrum_db(x, out=res)
    ts, cpu = [], []
    for _ in range(5):
        c0, t0 = time.process_time(), time.perf_counter(); pkg.spectrum_db(x, out=res); ts.append(time.perf_counter() - t0); cpu.append(time.process_time() - c0)
    ts.sort(); cpu.sort()
    print("  B=%%d: %%.2f ms median (%%.2f min) = %%.1f GB/s of input, cpu %%.0f ms" %% (b, ts[2] * 1e3, ts[0] * 1e3, x.nbytes / ts[2] / 1e9, cpu[2] * 1e3), flush=True)
''' % ROOT
for n in (sys.argv[1:] or ["7", "11", "15", "3"]):
    env = dict(os.environ)
    if n == "torch":                       # the same with PyTorch's bundled HIP runtime loaded first (what bench.py's process has)
        env["PROBE_TORCH_FIRST"] = "1"
    else:
        env["SDRK_HOST_THREADS"] = n
    print("SDRK_HOST_THREADS=%s" % n, flush=True)
    subprocess.run([sys.executable, "-c", CHILD], env=env, check=False)
