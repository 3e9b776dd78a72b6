#!/bin/bash
# round 3, GPU call 2: the rewritten per-row reductions: feature tests + timing, 3 vs 4 workgroups per CU
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r03b
mkdir -p "$OUT"
cd "$ROOT"
echo "== feature tests (base)"; timeout -k 10 300 python3 -m pytest tests/test_parity_gpu.py -m gpu -q -x -k "features or special_rows" > "$OUT/pytest_feat.log" 2>&1; tail -15 "$OUT/pytest_feat.log"
echo "== timing base (3 WG/CU)"; timeout -k 10 120 python3 tools/feat_probe.py 2>&1 | tee "$OUT/feat_base.log"
echo "== feature tests (w4)"; SDRK_LIB=$ROOT/sdr-iq-visualizer_amd/lib_w4/libsdrk.so timeout -k 10 300 python3 -m pytest tests/test_parity_gpu.py -m gpu -q -x -k "features or special_rows" > "$OUT/pytest_feat_w4.log" 2>&1; tail -3 "$OUT/pytest_feat_w4.log"
echo "== timing w4 (4 WG/CU, spills)"; SDRK_LIB=$ROOT/sdr-iq-visualizer_amd/lib_w4/libsdrk.so timeout -k 10 120 python3 tools/feat_probe.py 2>&1 | tee "$OUT/feat_w4.log"
