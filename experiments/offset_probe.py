#!/usr/bin/env python3
"""Round 6 probe (GPU box): does the level a read + write pair of the N = 4096 kernel runs at depend on an OFFSET of the output
inside one allocation (as it depends on which two allocations are paired, DESIGN_APPENDIX.md A.1)?  One 32 GiB input, one output
allocation of 16 GiB + 1 GiB; the plan's transform timed (warm, median of 7 launches) with the rows at offsets 0 ... 1 GiB.
    python3 experiments/offset_probe.py"""
import ctypes
import sys
import time

sys.path.insert(0, __file__.rsplit("/", 2)[0])
from sdr_iq_visualizer_amd import _ffi  # noqa: E402
from sdr_iq_visualizer_amd.spectrum import SpectrumPlan  # noqa: E402

lib = _ffi.lib()
n, nf = 4096, 1 << 20
pad = 1 << 30
for trial in range(2):
    d_in, d_out = ctypes.c_void_p(), ctypes.c_void_p()
    _ffi.check(lib.sdrk_dev_alloc(0, nf * n * 8, ctypes.byref(d_in)))
    _ffi.check(lib.sdrk_dev_alloc(0, nf * n * 4 + pad, ctypes.byref(d_out)))
    _ffi.check(lib.sdrk_synth_fill(0, 1234, 0, nf, n, d_in, None))
    with SpectrumPlan(n, window="hann") as p:
        t0 = time.perf_counter()
        while time.perf_counter() - t0 < 0.3:
            p.exec_device_timed(d_in.value, nf, d_out.value, 1)
        res = []
        for off in (0, 4096, 65536, 1 << 20, 2 << 20, 4 << 20, 8 << 20, 16 << 20, 32 << 20, 64 << 20, 96 << 20, 128 << 20, 256 << 20, 384 << 20,
                    512 << 20, 768 << 20, 1 << 30, 0):
            ms = sorted(p.exec_device_timed_each(d_in.value, nf, d_out.value + off, 7))
            res.append((off, ms[3]))
    print(f"allocation pair {trial}: " + "  ".join(f"{o >> 20 if o >= 1 << 20 else o / (1 << 20):g}M:{m:.3f}" for o, m in res), flush=True)
    lib.sdrk_dev_free(0, d_in)
    lib.sdrk_dev_free(0, d_out)
