"""Developer helper (GPU box): does what ran earlier in a process change the staged numpy boundary?  bench.py's
numpy_boundary leg (B = 32768, result reused) in fresh processes that first did nothing / imported torch / forked the
all-cores CPU leg / allocated and freed the bench's resident buffers / ran the large-frame legs.
python3 tools/boundary_context_probe.py"""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CHILD = r'''
import os, sys, time, ctypes
sys.path.insert(0, %r)
what = sys.argv[1].split("+")
import bench
if "fork" in what:
    bench.cpu_baseline_all_cores("hann", 1.0)
if "torch" in what:
    import torch
import numpy as np
import sdr_iq_visualizer_amd as pkg
from sdr_iq_visualizer_amd import _ffi, synth
from sdr_iq_visualizer_amd.spectrum import SpectrumPlan
lib = _ffi.lib()
if "resident" in what:
    plan = SpectrumPlan(4096, window="hann")
    d_in, d_out = ctypes.c_void_p(), ctypes.c_void_p()
    ms, chosen = (ctypes.c_float * 6)(), ctypes.c_int(0)
    _ffi.check(lib.sdrk_dev_alloc_stream_pair(0, (1 << 20) * 4096 * 8, (1 << 20) * 4096 * 4, 6, plan.handle, ctypes.byref(d_in), ctypes.byref(d_out), ms, ctypes.byref(chosen)))
    _ffi.check(lib.sdrk_synth_fill(0, 1234, 0, 1 << 20, 4096, d_in, None))
    plan.exec_device_timed_each(d_in.value, 1 << 20, d_out.value, launches=5)
    plan.close(); lib.sdrk_dev_free(0, d_in); lib.sdrk_dev_free(0, d_out)
if "large" in what:
    bench.device_config(lib, _ffi, SpectrumPlan, 0, 65536, 18749, 32768, "hann", 3)
    bench.device_config(lib, _ffi, SpectrumPlan, 0, 1 << 20, 256, 1 << 20, "hann", 3)
if "channel" in what:
    bench.channel_config5(lib, _ffi, pkg, SpectrumPlan, 0)
x = synth.synth_iq(1, 0, 32768, 4096)
if "small" in what:                               # the bench's earlier batch sizes through the same cached plan
    for b in (1, 16, 256, 4096):
        for _ in range(20): pkg.spectrum_db(x[0] if b == 1 else x[:b])
res = np.empty(x.shape, np.float32)
pkg.spectrum_db(x, out=res)
ts, cpu = [], []
for _ in range(5):
    c0, t0 = time.process_time(), time.perf_counter(); pkg.spectrum_db(x, out=res); ts.append(time.perf_counter() - t0); cpu.append(time.process_time() - c0)
ts.sort(); cpu.sort()
print("%%-34s B=32768 reused out: %%.2f ms median (%%.2f min) = %%.1f GB/s, cpu %%.0f ms" %% (sys.argv[1], ts[2] * 1e3, ts[0] * 1e3, x.nbytes / ts[2] / 1e9, cpu[2] * 1e3), flush=True)
''' % ROOT
for v in (sys.argv[1:] or ["clean", "torch", "fork+torch", "torch+resident", "torch+large", "torch+channel", "torch+small",
                           "fork+torch+resident+large+channel+small"]):
    subprocess.run([sys.executable, "-c", CHILD, v], check=False, stderr=subprocess.DEVNULL)
