#!/usr/bin/env python3
"""Developer experiment (GPU box): BASELINE configs 3 and 5 device-resident, the serial two-launch form against the
overlapped form (row pass of chunk i on a second stream beside the col pass of chunk i + 1) for several splits of
the CUs between the two roles.  Rows are compared bit for bit with the serial form's.
    python3 tools/overlap_probe.py [cfg3|cfg5|both]"""
import ctypes, json, os, statistics, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from sdr_iq_visualizer_amd import _ffi
from sdr_iq_visualizer_amd.spectrum import SpectrumPlan

lib = _ffi.lib()
dev = 0
which = sys.argv[1] if len(sys.argv) > 1 else "both"
cfgs = []
if which in ("cfg3", "both"):
    L = 614_400_000
    cfgs.append(("cfg3", 65536, 1 + (L - 65536) // 32768, 32768))
if which in ("cfg5", "both"):
    cfgs.append(("cfg5", 1 << 20, 256, 1 << 20))
splits = [None, (128, 128), (160, 96), (96, 160), (192, 128), (128, 192), (192, 192), (256, 256), (171, 128), (144, 112)]

for name, nfft, n_frames, stride in cfgs:
    in_samples = (n_frames - 1) * stride + nfft
    gen_frames = (in_samples + 4095) // 4096
    d_gen, d_out = ctypes.c_void_p(), ctypes.c_void_p()
    _ffi.check(lib.sdrk_dev_alloc(dev, gen_frames * 4096 * 8, ctypes.byref(d_gen)))
    _ffi.check(lib.sdrk_dev_alloc(dev, n_frames * nfft * 4, ctypes.byref(d_out)))
    _ffi.check(lib.sdrk_synth_fill(dev, 99, 0, gen_frames, 4096, d_gen, None))
    algo = 8 * in_samples + 4 * n_frames * nfft
    ref = None
    # rows sampled for the bit comparison: first, last, and a stride through the middle
    pick = sorted(set([0, 1, n_frames - 1] + list(range(0, n_frames, max(1, n_frames // 61)))))
    for sp in splits:
        for k in ("SDRK_OVL_COL_CUS", "SDRK_OVL_ROW_CUS"):
            os.environ.pop(k, None)
        if sp is not None:
            os.environ["SDRK_OVL_COL_CUS"], os.environ["SDRK_OVL_ROW_CUS"] = str(sp[0]), str(sp[1])
        with SpectrumPlan(nfft, window="hann", device=dev, overlap_passes=sp is not None) as plan:
            plan.exec_device(d_gen.value, n_frames, d_out.value, frame_stride=stride)
            plan.sync()
            ms = plan.exec_device_timed_each(d_gen.value, n_frames, d_out.value, 9, frame_stride=stride)
            rows = np.empty((len(pick), nfft), dtype=np.float32)
            for i, f in enumerate(pick):
                _ffi.check(lib.sdrk_memcpy_d2h(dev, rows[i].ctypes.data_as(ctypes.c_void_p),
                                               ctypes.c_void_p(d_out.value + f * nfft * 4), nfft * 4))
        if ref is None:
            ref = rows
        same = bool(np.array_equal(rows.view(np.uint32), ref.view(np.uint32)))
        t = statistics.median(ms)
        print(json.dumps({"cfg": name, "split_col_row_cus": sp, "ms": round(t, 4), "min": round(min(ms), 4), "max": round(max(ms), 4),
                          "frac": round(algo / (t * 1e-3) / 1e9 / 8000, 4), "bit_identical_to_serial": same}), flush=True)
    lib.sdrk_dev_free(dev, d_out)
    lib.sdrk_dev_free(dev, d_gen)
