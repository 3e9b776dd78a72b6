#!/bin/bash
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r03k
mkdir -p "$OUT"
cd "$ROOT"
echo "== tests"; timeout -k 10 600 python3 -m pytest tests/test_parity_gpu.py -m gpu -q -x -k "waterfall or channel" > "$OUT/pytest.log" 2>&1; tail -3 "$OUT/pytest.log"
echo "== channel leg"; timeout -k 10 300 python3 - <<'PY' 2>&1 | tee "$OUT/channel.json"
import json, sys
sys.path.insert(0, ".")
import bench
import sdr_iq_visualizer_amd as pkg
from sdr_iq_visualizer_amd import _ffi
from sdr_iq_visualizer_amd.spectrum import SpectrumPlan
print(json.dumps(bench.channel_config5(_ffi.lib(), _ffi, pkg, SpectrumPlan, 0)))
PY
