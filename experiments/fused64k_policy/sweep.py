#!/usr/bin/env python3
"""Round-6 gated experiment (GPU box): can the fused N = 65536 kernel's ring stay inside the XCD's L2?

Times fused64k_kernel (library built by build.sh, selected with SDRK_LIB) for every point of
    workgroups per CU {3, 2, 1}  (= ring sets per XCD; x ring depth x 512 KiB of a 4 MiB L2)
  x cache policy of the streamed input loads   (buffer aux bits: 2 = nt, 0 = default, 18 = sc1 nt, 19 = sc0 sc1 nt)
  x cache policy of the streamed row stores    (2 = nt, 0 = default, 16 = sc1, 17 = sc0 sc1, 19 = sc0 sc1 nt)
  x {no-wait timing build, real build}
warm (by time), transforms back to back with an event between consecutive ones, and holds the real builds' rows against the
two tiled launches' (bit-identical or not).  One JSON line per point; a table at the end.

    SDRK_LIB=.../lib_fuexp/libsdrk.so python3 experiments/fused64k_policy/sweep.py [frames] [hop] [--quick]
"""
import ctypes
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from sdr_iq_visualizer_amd import _ffi  # noqa: E402
from sdr_iq_visualizer_amd.spectrum import SpectrumPlan  # noqa: E402

args = [a for a in sys.argv[1:] if not a.startswith("--")]
quick = "--quick" in sys.argv
nowait_only = "--nowait-only" in sys.argv   # libraries whose real build is meaningless (FU_ALIAS_RING)
n = 65536
nf = int(args[0]) if len(args) > 0 else 4096
hop = int(args[1]) if len(args) > 1 else n
lib = _ffi.lib()
samples = (nf - 1) * hop + n
d_in, d_a, d_b = ctypes.c_void_p(), ctypes.c_void_p(), ctypes.c_void_p()
_ffi.check(lib.sdrk_dev_alloc(0, samples * 8, ctypes.byref(d_in)))
_ffi.check(lib.sdrk_dev_alloc(0, nf * n * 4, ctypes.byref(d_a)))
_ffi.check(lib.sdrk_dev_alloc(0, nf * n * 4, ctypes.byref(d_b)))
_ffi.check(lib.sdrk_synth_fill(0, 3, 0, (samples + 4095) // 4096, 4096, d_in, None))


def timed(plan, d_out, warm_ms=60.0, k=12):
    t0 = time.perf_counter()
    w = 0
    while (time.perf_counter() - t0) * 1e3 < warm_ms or w < 2:
        plan.exec_device_timed(d_in.value, nf, d_out.value, 1, frame_stride=hop)
        w += 1
    ms = sorted(plan.exec_device_timed_each(d_in.value, nf, d_out.value, k, frame_stride=hop))
    return ms[len(ms) // 2], ms[0]


rows = min(nf, 48)
ref = []
with SpectrumPlan(n, window="hann", fused64k=False) as p:
    med, mn = timed(p, d_a)
    print(json.dumps({"kernel": "tiled (two launches)", "frames": nf, "hop": hop, "median_ms": round(med, 4), "min_ms": round(mn, 4),
                      "library": _ffi.library_path()}), flush=True)
    tiled_ms = med
    for off in (0, (nf - rows) * n * 4, (nf // 2) * n * 4):
        a = np.empty(rows * n, np.float32)
        _ffi.check(lib.sdrk_memcpy_d2h(0, a.ctypes.data_as(ctypes.c_void_p), ctypes.c_void_p(d_a.value + off), a.nbytes))
        ref.append((off, a))

ins = (2, 0, 18, 19)
outs = (2, 0, 16, 17, 19)
if quick:
    ins, outs = (2, 0), (2, 17)
if "--few" in sys.argv:      # libraries built with -DFU_FEW
    ins, outs = (2,), (2, 17)
table = []
dead = set()
for wgpc in (3, 2, 1):
    for pin in ins:
        for pout in outs:
            for nowait in ((1,) if nowait_only else (1, 0)):
                if wgpc in dead:
                    continue
                os.environ.update(SDRK_FU_WG_PER_CU=str(wgpc), SDRK_FU_IN_AUX=str(pin), SDRK_FU_OUT_AUX=str(pout),
                                  SDRK_FU_NOWAIT=str(nowait))
                rec = {"wg_per_cu": wgpc, "in_aux": pin, "out_aux": pout, "nowait": nowait}
                try:
                    with SpectrumPlan(n, window="hann", fused64k=True) as p:
                        p.exec_device(d_in.value, nf, d_b.value, frame_stride=hop)
                        p.sync()
                        med, mn = timed(p, d_b)
                        p.sync()
                        rec.update(median_ms=round(med, 4), min_ms=round(mn, 4), vs_tiled=round(med / tiled_ms, 3))
                        if not nowait:
                            same = True
                            for off, a in ref:
                                b = np.empty_like(a)
                                _ffi.check(lib.sdrk_memcpy_d2h(0, b.ctypes.data_as(ctypes.c_void_p), ctypes.c_void_p(d_b.value + off), b.nbytes))
                                same = same and bool(np.array_equal(a, b))
                            rec["rows_identical_to_tiled"] = same
                except Exception as e:  # a set that never formed: the launch reports it; give this wg_per_cu up
                    rec["error"] = str(e)[:200]
                    dead.add(wgpc)
                print(json.dumps(rec), flush=True)
                table.append(rec)

print(f"\n# fused64k policy sweep: {nf} frames, hop {hop}; tiled two-pass {tiled_ms:.3f} ms; ring depth from the library build")
print("# wg/CU in_aux out_aux | no-wait ms | real ms | real/tiled | identical")
key = lambda r: (r["wg_per_cu"], r["in_aux"], r["out_aux"])
by = {}
for r in table:
    by.setdefault(key(r), {})[r["nowait"]] = r
for k, v in by.items():
    nw, re_ = v.get(1, {}), v.get(0, {})
    print(f"  {k[0]}     {k[1]:>3}    {k[2]:>3}    | {nw.get('median_ms', 'err'):>8} | {re_.get('median_ms', 'err'):>8} | "
          f"{re_.get('vs_tiled', '-'):>6} | {re_.get('rows_identical_to_tiled', '-')}")
for d in (d_in, d_a, d_b):
    lib.sdrk_dev_free(0, d)
