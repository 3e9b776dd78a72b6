#!/bin/bash
# round 3, GPU call 6: pinned host path tests + numpy-boundary leg + pcie probe (register cost) + bench --gpus 2 rehearsal
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r03f
mkdir -p "$OUT" "$ROOT/sdr-iq-visualizer_amd/build_tools"
cd "$ROOT"
echo "== pinned test"; timeout -k 10 600 python3 -m pytest tests/test_parity_gpu.py -m gpu -q -x -k "pinned or sharding or thread_per_device or host" > "$OUT/pytest_pinned.log" 2>&1; tail -5 "$OUT/pytest_pinned.log"
echo "== pcie probe"; hipcc --offload-arch=gfx950 -O3 -o sdr-iq-visualizer_amd/build_tools/pcie_probe sdr-iq-visualizer_amd/csrc/tools/pcie_probe.hip 2>/dev/null && timeout -k 10 120 ./sdr-iq-visualizer_amd/build_tools/pcie_probe 2>&1 | tee "$OUT/pcie_probe.log"
echo "== numpy boundary"; timeout -k 10 300 python3 - <<'PY' 2>&1 | tee "$OUT/numpy_boundary.json"
import json, sys
sys.path.insert(0, ".")
import bench, ctypes
import sdr_iq_visualizer_amd as pkg
from sdr_iq_visualizer_amd import _ffi, synth
print(json.dumps(bench.numpy_boundary(_ffi.lib(), _ffi, pkg, synth, 0)))
PY
echo "== bench --gpus 2 rehearsal (self-launched, gloo, ranks share the GPU)"; timeout -k 10 500 python3 bench.py --gpus 2 --steps 3 --warmup 1 --frames 262144 --cpu-seconds 3 --cpu-all-cores-seconds 2 > "$OUT/bench_gpus2.json" 2> "$OUT/bench_gpus2.err"; echo rc=$?; tail -c 1500 "$OUT/bench_gpus2.json"; tail -5 "$OUT/bench_gpus2.err"
