#!/bin/bash
# round 3, GPU call 7: the single-pass N = 32768 kernel: parity, then timing against the two-pass form
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r03h
mkdir -p "$OUT"
cd "$ROOT"
echo "== parity (new kernel)"; timeout -k 10 300 python3 - <<'PY' 2>&1 | tee "$OUT/parity32k.log"
import numpy as np, sys
sys.path.insert(0, ".")
import sdr_iq_visualizer_amd as pkg
from oracle import cpu_ref
from tests.parity import assert_db_parity, assert_complex_parity
rng = np.random.default_rng(1)
n = 32768
for b in (1, 3, 300):
    x = ((rng.standard_normal((b, n)) + 1j * rng.standard_normal((b, n))) * 3).astype(np.complex64)
    assert_db_parity(pkg.spectrum_db(x), cpu_ref.spectrum_db(x), what=f"rect b={b}")
    assert_db_parity(pkg.spectrum_db(x, window="hann"), cpu_ref.spectrum_db(x, window=np.hanning(n)), what=f"hann b={b}")
    assert_db_parity(pkg.spectrum_db(x, shift=False), cpu_ref.spectrum_db(x, shift=False), what="noshift")
    assert_complex_parity(pkg.fft_c64(x), cpu_ref.fft(x), what="fft")
    assert_complex_parity(pkg.fft_c64(x, window="hann", shift=True), cpu_ref.fft(x, window=np.hanning(n), shift=True), what="fft hann shift")
s = ((rng.standard_normal(20 * 8192 + n) + 1j * rng.standard_normal(20 * 8192 + n))).astype(np.complex64)
assert_db_parity(pkg.stft_db(s, n, 8192, "hann"), cpu_ref.stft_db(s, n, 8192, window=cpu_ref.hann(n)), what="stft 75% overlap")
print("parity ok")
PY
for v in base t32; do
  lib=$ROOT/sdr-iq-visualizer_amd/lib_$v/libsdrk.so; [ "$v" = base ] && lib=$ROOT/sdr-iq-visualizer_amd/lib/libsdrk.so
  for w in rect hann; do echo -n "$v $w: "; SDRK_LIB=$lib python3 tools/one_config.py 32768 4096 32768 $w | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms'], 'ms', d['algorithmic_GBps'], 'GB/s')"; done
done 2>&1 | tee "$OUT/timing32k.log"
echo "== tests touching 32768"; timeout -k 10 600 python3 -m pytest tests/test_parity_gpu.py -m gpu -q -x -k "every_power or windows or stft or sharding or randomised or pinned" > "$OUT/pytest.log" 2>&1; tail -3 "$OUT/pytest.log"
