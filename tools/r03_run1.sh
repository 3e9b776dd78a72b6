#!/bin/bash
# round 3, first GPU call: new parity tests, overlap sweep, f1 kernel profile
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r03a
mkdir -p "$OUT"
cd "$ROOT"
echo "== new parity tests"; python3 -m pytest tests/test_parity_gpu.py -m gpu -q -x -k "golden" -s > "$OUT/pytest_golden.log" 2>&1; tail -5 "$OUT/pytest_golden.log"
echo "== overlap sweep"; timeout -k 10 400 python3 tools/overlap_probe.py both > "$OUT/overlap_probe.log" 2>&1; cat "$OUT/overlap_probe.log"
echo "== f1 profile"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/feat_trace" -- python3 "$ROOT/tools/feat_probe.py" > "$OUT/feat_trace.stdout" 2> "$OUT/feat_trace.stderr"
cat "$OUT/feat_trace.stdout"
cat "$OUT"/feat_trace/*/*kernel_stats.csv | cut -c1-200
cd "$ROOT"
echo "== bench short"; python3 bench.py --no-secondary --cpu-seconds 0 --steps 10 > "$OUT/bench_short.json" 2> "$OUT/bench_short.err"; python3 -c "
import json; l=json.load(open('$OUT/bench_short.json')); print(l['value'], l['roofline'])"
