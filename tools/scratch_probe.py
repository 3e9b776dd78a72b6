"""Developer helper (GPU box): does WHICH scratch allocation a large-frame plan got change its speed?
K plans of the same length (each with its own scratch), the same input / output buffers, median of 7 launches each;
then the same plans again (stability)."""
import ctypes, sys
sys.path.insert(0, __file__.rsplit("/", 2)[0])
from sdr_iq_visualizer_amd import _ffi
from sdr_iq_visualizer_amd.spectrum import SpectrumPlan
n = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
nf = int(sys.argv[2]) if len(sys.argv) > 2 else 18749
hop = int(sys.argv[3]) if len(sys.argv) > 3 else n // 2
K = int(sys.argv[4]) if len(sys.argv) > 4 else 6
lib = _ffi.lib()
samples = (nf - 1) * hop + n
d_in, d_out = ctypes.c_void_p(), ctypes.c_void_p()
_ffi.check(lib.sdrk_dev_alloc(0, ((samples + 4095) // 4096) * 4096 * 8, ctypes.byref(d_in)))
_ffi.check(lib.sdrk_dev_alloc(0, nf * n * 4, ctypes.byref(d_out)))
_ffi.check(lib.sdrk_synth_fill(0, 3, 0, (samples + 4095) // 4096, 4096, d_in, None))
plans = [SpectrumPlan(n, window="hann") for _ in range(K)]
for rnd in range(2):
    out = []
    for p in plans:
        p.exec_device(d_in.value, nf, d_out.value, frame_stride=hop); p.sync()
        ms = sorted(p.exec_device_timed_each(d_in.value, nf, d_out.value, 7, frame_stride=hop))
        out.append(ms[3])
    print(f"N={n} frames={nf} hop={hop} round {rnd}: " + " ".join(f"{v:.3f}" for v in out), flush=True)
