#!/bin/bash
# Everything measured for a round, in one gpurun call (run from the repo root on the GPU box):
#   tools/collect_round.sh r02
# -> gpurun_out/collect_<tag>/ : rocprofv3 evidence (tools/profile_round.sh), the bench line, the size sweep,
#    the placement probes and N bench processes with and without placement tuning.
TAG=${1:-r02}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/collect_$TAG
mkdir -p "$OUT"
cd "$ROOT"
echo "== bench (full line)"; python3 bench.py > "$OUT/bench_n1.json" 2> "$OUT/bench_n1.err"
echo "== profiles"; bash tools/profile_round.sh $TAG > "$OUT/profile_round.log" 2>&1; cp gpurun_out/prof_$TAG/summary.json "$OUT/rocprof_summary.json"; cp gpurun_out/prof_$TAG/pmc_fetch_write_rows.csv "$OUT/" 2>/dev/null
for t in bench cfg3 cfg5; do cp gpurun_out/prof_$TAG/${t}_trace/*/*kernel_stats.csv "$OUT/${t}_kernel_stats.csv" 2>/dev/null; done
echo "== size sweep"; python3 tools/size_sweep.py > "$OUT/size_sweep.log" 2>&1
echo "== placement probes"
mkdir -p sdr-iq-visualizer_amd/build_tools
for t in placeprobe queueprobe; do hipcc --offload-arch=gfx950 -O3 -o sdr-iq-visualizer_amd/build_tools/$t sdr-iq-visualizer_amd/csrc/tools/$t.hip 2>/dev/null; done
{ for i in 1 2 3; do ./sdr-iq-visualizer_amd/build_tools/placeprobe 20 3; done; for i in 1 2 3 4; do ./sdr-iq-visualizer_amd/build_tools/queueprobe 19 4; done; } > "$OUT/placeprobe.log" 2>&1
echo "== 16 bench processes with placement tuning, 8 without"
summ() { python3 -c "
import json,sys
l=json.loads(sys.stdin.read()); t=l['telemetry']
print('%.4f  %.4f  %.4f  %.4f  %7.1f  %.4f  sclk_after %s  probe_ms %s chosen %d' % (l['launch_ms']['median'], l['launch_ms']['min'], l['launch_ms']['max'], l['roofline']['frac'], l['roofline']['measured_copy_GBps'], l['roofline']['frac_of_measured_copy'], t['after']['sclk_mhz'], l['placement']['probe_ms'], l['placement']['chosen']))"; }
{ echo "# python bench.py --no-secondary --cpu-seconds 0 --parity-frames 0   (16 processes; placement: 6 candidates)";
  echo "# launch_ms median  min  max  frac_of_8TB/s  copy_GB/s  kernel/copy";
  for i in $(seq 16); do python3 bench.py --no-secondary --cpu-seconds 0 --parity-frames 0 2>/dev/null | tail -1 | summ; done; } > "$OUT/placement_16_runs.txt"
{ echo "# the same with --placement-candidates 1 (plain alloc(in); alloc(out)), 8 processes";
  for i in $(seq 8); do python3 bench.py --no-secondary --cpu-seconds 0 --parity-frames 0 --placement-candidates 1 2>/dev/null | tail -1 | summ; done; } > "$OUT/plain_alloc_8_runs.txt"
echo "== fused vs tiled N=65536 (4096 packed frames, then config 3; Hann)"
{ python3 tools/fused_probe.py 4096 65536 hann; python3 tools/fused_probe.py 18749 32768 hann; } > "$OUT/fused64k_vs_tiled.log" 2>&1
cat "$OUT/fused64k_vs_tiled.log"; tail -3 "$OUT/placement_16_runs.txt"; tail -3 "$OUT/plain_alloc_8_runs.txt"
echo "== one-frame host call"; python3 tools/small_call_probe.py > "$OUT/small_call_probe.log" 2>&1; cat "$OUT/small_call_probe.log"
echo "== gpu tests"; python3 -m pytest tests -m gpu -q > "$OUT/pytest_gpu.log" 2>&1; tail -2 "$OUT/pytest_gpu.log"
