#!/bin/bash
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r03j
mkdir -p "$OUT"
cd "$ROOT"
echo "== tests"; timeout -k 10 600 python3 -m pytest tests/test_parity_gpu.py -m gpu -q -x -k "pinned or randomised or peak_scan or features" > "$OUT/pytest.log" 2>&1; tail -3 "$OUT/pytest.log"
echo "== numpy boundary"; timeout -k 10 300 python3 - <<'PY' 2>&1 | tee "$OUT/numpy_boundary.json"
import json, sys
sys.path.insert(0, ".")
import bench
import sdr_iq_visualizer_amd as pkg
from sdr_iq_visualizer_amd import _ffi, synth
r = bench.numpy_boundary(_ffi.lib(), _ffi, pkg, synth, 0)
for k, v in r["by_batch"].items(): print(k, v)
PY
