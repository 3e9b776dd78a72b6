#!/bin/bash
# round 5: the bench-related part of the collection again after bench.py's default went from 6 to 10 placement candidates
# (bench line, its kernel trace + FETCH / WRITE passes, the tuned / plain bench processes, the GPU suite)
TAG=r05
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/collect_$TAG; P=$ROOT/gpurun_out/prof_$TAG
mkdir -p "$OUT" "$P"
cd "$ROOT"
echo "== bench (full line)"; python3 bench.py > "$OUT/bench_n1.json" 2> "$OUT/bench_n1.err"; tail -c 300 "$OUT/bench_n1.err"
( cd /tmp && export TMPDIR=/tmp
  BENCH_SHORT="--steps 3 --warmup 1 --cpu-seconds 0 --parity-frames 0 --no-secondary"
  rm -rf "$P/bench_trace" "$P/bench_fetch" "$P/bench_write"
  rocprofv3 --kernel-trace --stats --output-format csv -d "$P/bench_trace" -- python3 "$ROOT/bench.py" --no-secondary --cpu-seconds 0 > "$P/bench_trace.stdout" 2> "$P/bench_trace.stderr"
  rocprofv3 --pmc FETCH_SIZE --output-format csv -d "$P/bench_fetch" -- python3 "$ROOT/bench.py" $BENCH_SHORT > "$P/bench_fetch.stdout" 2> "$P/bench_fetch.stderr"
  rocprofv3 --pmc WRITE_SIZE --output-format csv -d "$P/bench_write" -- python3 "$ROOT/bench.py" $BENCH_SHORT > "$P/bench_write.stdout" 2> "$P/bench_write.stderr" )
( cd /tmp && export TMPDIR=/tmp    # the per-row reductions again (they changed after the first take): trace + the SQ counter sets
  for t in feat_trace feat_sq1 feat_sq2 feat_sq3 feat_sq4; do rm -rf "$P/$t"; done
  rocprofv3 --kernel-trace --stats --output-format csv -d "$P/feat_trace" -- python3 "$ROOT/tools/feat_probe.py" > "$P/feat_trace.stdout" 2> "$P/feat_trace.stderr"
  rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU --output-format csv -d "$P/feat_sq1" -- python3 "$ROOT/tools/feat_probe.py" > /dev/null 2> "$P/feat_sq1.stderr"
  rocprofv3 --pmc SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM SQ_INSTS_SMEM --output-format csv -d "$P/feat_sq2" -- python3 "$ROOT/tools/feat_probe.py" > /dev/null 2> "$P/feat_sq2.stderr"
  rocprofv3 --pmc SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_ANY --output-format csv -d "$P/feat_sq3" -- python3 "$ROOT/tools/feat_probe.py" > /dev/null 2> "$P/feat_sq3.stderr"
  rocprofv3 --pmc SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_WAIT_ANY SQ_LDS_BANK_CONFLICT --output-format csv -d "$P/feat_sq4" -- python3 "$ROOT/tools/feat_probe.py" > /dev/null 2> "$P/feat_sq4.stderr" )
cp "$P"/feat_trace/*/*kernel_stats.csv "$OUT/feat_kernel_stats.csv" 2>/dev/null
python3 tools/summarise_profiles.py "$P" > "$P/summary.json" 2> "$P/summary.err"; cp "$P/summary.json" "$OUT/rocprof_summary.json"
python3 tools/summarise_profiles.py "$P" --rows "$OUT/pmc_fetch_write_rows.csv" 2>> "$P/summary.err"
cp "$P"/bench_trace/*/*kernel_stats.csv "$OUT/bench_kernel_stats.csv" 2>/dev/null
summ() { python3 -c "
import json,sys
l=json.loads(sys.stdin.read()); t=l['telemetry']
print('%.4f  %.4f  %.4f  %.4f  %7.1f  %.4f  sclk_after %s  probe_ms %s chosen %d' % (l['launch_ms']['median'], l['launch_ms']['min'], l['launch_ms']['max'], l['roofline']['frac'], l['roofline']['measured_copy_GBps'], l['roofline']['frac_of_measured_copy'], t['after']['sclk_mhz'], l['placement']['probe_ms'], l['placement']['chosen']))"; }
{ echo "# python bench.py --no-secondary --cpu-seconds 0 --parity-frames 0   (6 processes; placement: 10 candidates)";
  echo "# launch_ms median  min  max  frac_of_8TB/s  copy_GB/s  kernel/copy";
  for i in 1 2 3 4 5 6; do python3 bench.py --no-secondary --cpu-seconds 0 --parity-frames 0 2>/dev/null | tail -1 | summ; done; } > "$OUT/placement_runs.txt"
{ echo "# the same with --placement-candidates 1 (plain alloc(in); alloc(out)), 3 processes";
  for i in 1 2 3; do python3 bench.py --no-secondary --cpu-seconds 0 --parity-frames 0 --placement-candidates 1 2>/dev/null | tail -1 | summ; done; } > "$OUT/plain_alloc_runs.txt"
cat "$OUT/placement_runs.txt" "$OUT/plain_alloc_runs.txt"
echo "== gpu tests"; python3 -m pytest tests -m gpu -q > "$OUT/pytest_gpu.log" 2>&1; tail -2 "$OUT/pytest_gpu.log"
