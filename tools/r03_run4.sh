#!/bin/bash
# round 3, GPU call 4: full GPU suite + stand-alone reduction kernel occupancy A/B + kernel trace of the overlap experiment
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r03d
mkdir -p "$OUT"
cd "$ROOT"
for v in base rf2 rf4; do
  lib=$ROOT/sdr-iq-visualizer_amd/lib_$v/libsdrk.so; [ "$v" = base ] && lib=$ROOT/sdr-iq-visualizer_amd/lib/libsdrk.so
  echo "== feat_probe $v"; SDRK_LIB=$lib timeout -k 10 120 python3 tools/feat_probe.py 2>&1 | tee "$OUT/feat_$v.log"
done
echo "== full gpu suite"; timeout -k 10 900 python3 -m pytest tests -m gpu -q -x > "$OUT/pytest_gpu.log" 2>&1; tail -5 "$OUT/pytest_gpu.log"
