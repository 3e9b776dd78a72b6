#!/bin/bash
# round 3, GPU call 5: overlap flag test, row-pass spill fix timing, kernel traces of the overlap experiment
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r03e
mkdir -p "$OUT"
cd "$ROOT"
echo "== tests"; timeout -k 10 600 python3 -m pytest tests/test_parity_gpu.py -m gpu -q -x -k "overlapped or largest or every_power or golden_large or stft or config3 or fused_n65536" > "$OUT/pytest_large.log" 2>&1; tail -3 "$OUT/pytest_large.log"
echo "== configs 3 and 5 (serial form, three processes each)"
for i in 1 2 3; do python3 tools/one_config.py 65536 18749 32768 hann; python3 tools/one_config.py 1048576 256 1048576 hann; done 2>&1 | python3 -c "
import sys,json
for l in sys.stdin:
    try: d=json.loads(l)
    except Exception: print(l.strip()); continue
    print(d['nfft'], d['ms'], 'ms', d.get('algorithmic_GBps'), 'GB/s')" | tee "$OUT/cfg35.log"
echo "== overlap sweep"; timeout -k 10 400 python3 tools/overlap_probe.py both > "$OUT/overlap_probe.log" 2>&1; tail -25 "$OUT/overlap_probe.log"
cd /tmp && export TMPDIR=/tmp
for cfg in cfg3 cfg5; do for mode in serial overlap; do
  rocprofv3 --kernel-trace --output-format csv -d "$OUT/trace_${cfg}_$mode" -- python3 "$ROOT/tools/overlap_trace.py" $cfg $mode > "$OUT/trace_${cfg}_$mode.stdout" 2> "$OUT/trace_${cfg}_$mode.stderr"
  cat "$OUT/trace_${cfg}_$mode.stdout"
  python3 "$ROOT/tools/summarise_overlap_trace.py" "$OUT/trace_${cfg}_$mode" "$cfg $mode"
done; done 2>&1 | tee "$OUT/overlap_trace_summary.txt"
cd "$ROOT"
cp "$OUT"/trace_cfg3_overlap/*/*kernel_trace.csv "$OUT/cfg3_overlap_kernel_trace.csv" 2>/dev/null
rm -rf "$OUT"/trace_*/
