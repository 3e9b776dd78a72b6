#!/usr/bin/env python3
"""Developer helper (GPU box, under rocprofv3 --kernel-trace): BASELINE config 3 or 5 device-resident, three
launches, serial or overlapped form of the two passes.
    python3 tools/overlap_trace.py cfg3|cfg5 serial|overlap"""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sdr_iq_visualizer_amd import _ffi
from sdr_iq_visualizer_amd.spectrum import SpectrumPlan

lib, dev = _ffi.lib(), 0
cfg, mode = sys.argv[1], sys.argv[2]
nfft, n_frames, stride = (65536, 1 + (614_400_000 - 65536) // 32768, 32768) if cfg == "cfg3" else (1 << 20, 256, 1 << 20)
in_samples = (n_frames - 1) * stride + nfft
gen = (in_samples + 4095) // 4096
d_in, d_out = ctypes.c_void_p(), ctypes.c_void_p()
_ffi.check(lib.sdrk_dev_alloc(dev, gen * 4096 * 8, ctypes.byref(d_in)))
_ffi.check(lib.sdrk_dev_alloc(dev, n_frames * nfft * 4, ctypes.byref(d_out)))
_ffi.check(lib.sdrk_synth_fill(dev, 99, 0, gen, 4096, d_in, None))
with SpectrumPlan(nfft, window="hann", device=dev, overlap_passes=(mode == "overlap")) as plan:
    for _ in range(3):
        plan.exec_device(d_in.value, n_frames, d_out.value, frame_stride=stride)
        plan.sync()
    ms = plan.exec_device_timed_each(d_in.value, n_frames, d_out.value, 3, frame_stride=stride)
print(cfg, mode, "launch ms (under the profiler):", [round(v, 3) for v in ms])
