import sys, time, os
sys.path.insert(0, '.')
import numpy as np
import sdr_iq_visualizer_amd as pkg
from oracle import cpu_ref
from sdr_iq_visualizer_amd import synth
from sdr_iq_visualizer_amd.spectrum import SpectrumPlan
for nf in (1, 2, 7, 40, 300):
    x = synth.synth_iq(5, 0, nf * 16, 4096).reshape(nf, 65536)
    t = time.time()
    try:
        with SpectrumPlan(65536) as p:
            got = p.spectrum_db(x)
        ref = cpu_ref.spectrum_db(x)
        mg, mr = 10.0 ** (got.astype(np.float64) / 20), 10.0 ** (ref.astype(np.float64) / 20)
        print(nf, "frames: err", float((np.abs(mg - mr) / mr.max(axis=-1, keepdims=True)).max()), "time %.2fs" % (time.time() - t), flush=True)
    except Exception as e:
        print(nf, "frames: EXC", e, "time %.2fs" % (time.time() - t), flush=True)
        break
