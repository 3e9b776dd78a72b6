#!/bin/bash
# Everything measured for a round, in one gpurun call (run from the repo root on the GPU box):
#   tools/collect_round.sh r02
# -> gpurun_out/collect_<tag>/ : rocprofv3 evidence (tools/profile_round.sh), the bench line, the size sweep,
#    the placement probes and N bench processes with and without placement tuning.
# SDRK_COLLECT_LIGHT=1: what changed or is quoted this round only — skips the placement micro-probes and the overlap A/B
# runs (closed experiments: their logs of the round that ran them stand) and halves the bench-process counts.
TAG=${1:-r06}
LIGHT=${SDRK_COLLECT_LIGHT:-0}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/collect_$TAG
mkdir -p "$OUT"
cd "$ROOT"
# SDRK_COLLECT_PART=A: the bench line, the rocprofv3 evidence, the channel trace, the size sweep; =B: everything after that.
# (One gpurun call is limited to 20 minutes; the two parts are two calls.  Unset: both.)
PART=${SDRK_COLLECT_PART:-AB}
case $PART in *A*)
echo "== bench (full line)"; python3 bench.py > "$OUT/bench_n1.json" 2> "$OUT/bench_n1.err"
echo "== profiles"; bash tools/profile_round.sh $TAG > "$OUT/profile_round.log" 2>&1; cp gpurun_out/prof_$TAG/summary.json "$OUT/rocprof_summary.json"; cp gpurun_out/prof_$TAG/pmc_fetch_write_rows.csv "$OUT/" 2>/dev/null
for t in bench cfg3 cfg3tiled cfg5 n16384 n2p21 n2p22 feat; do cp gpurun_out/prof_$TAG/${t}_trace/*/*kernel_stats.csv "$OUT/${t}_kernel_stats.csv" 2>/dev/null; done
cp gpurun_out/prof_$TAG/cfg3_trace_vs_events.json gpurun_out/prof_$TAG/cfg3tiled_trace_vs_events.json gpurun_out/prof_$TAG/cfg5_trace_vs_events.json gpurun_out/prof_$TAG/cfg3_steady.json gpurun_out/prof_$TAG/cfg3tiled_steady.json gpurun_out/prof_$TAG/cfg5_steady.json "$OUT/" 2>/dev/null
echo "== the continuous channel (config 5 as worded), kernel-traced"
( cd /tmp && export TMPDIR=/tmp && rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/channel_trace" -- python3 "$ROOT/tools/channel_probe.py" --batch 24 > "$OUT/channel_probe.json" 2> /dev/null )
python3 tools/summarise_channel_trace.py "$OUT/channel_trace" --batches 11 --runs 6 > "$OUT/channel_trace_summary.json" 2>/dev/null; cp "$OUT"/channel_trace/*/*kernel_stats.csv "$OUT/channel_kernel_stats.csv" 2>/dev/null; rm -rf "$OUT/channel_trace"
echo "== size sweep"; python3 tools/size_sweep.py > "$OUT/size_sweep.log" 2>&1
;; esac
case $PART in *B*) ;; *) exit 0;; esac
if [ "$LIGHT" != 1 ]; then
echo "== placement probes"
mkdir -p sdr-iq-visualizer_amd/build_tools
for t in placeprobe queueprobe; do hipcc --offload-arch=gfx950 -O3 -o sdr-iq-visualizer_amd/build_tools/$t experiments/probes/$t.hip 2>/dev/null; done
{ for i in 1 2 3; do ./sdr-iq-visualizer_amd/build_tools/placeprobe 20 3; done; for i in 1 2 3 4; do ./sdr-iq-visualizer_amd/build_tools/queueprobe 19 4; done; } > "$OUT/placeprobe.log" 2>&1
fi
NP=12; NQ=6; [ "$LIGHT" = 1 ] && { NP=6; NQ=3; }
echo "== $NP bench processes with placement tuning, $NQ without"
summ() { python3 -c "
import json,sys
l=json.loads(sys.stdin.read()); t=l['telemetry']
print('%.4f  %.4f  %.4f  %.4f  %7.1f  %.4f  sclk_after %s  probe_ms %s chosen %d' % (l['launch_ms']['median'], l['launch_ms']['min'], l['launch_ms']['max'], l['roofline']['frac'], l['roofline']['measured_copy_GBps'], l['roofline']['frac_of_measured_copy'], t['after']['sclk_mhz'], l['placement']['probe_ms'], l['placement']['chosen']))"; }
{ echo "# python bench.py --no-secondary --cpu-seconds 0 --parity-frames 0   ($NP processes; placement: 6 candidates)";
  echo "# launch_ms median  min  max  frac_of_8TB/s  copy_GB/s  kernel/copy";
  for i in $(seq $NP); do python3 bench.py --no-secondary --cpu-seconds 0 --parity-frames 0 2>/dev/null | tail -1 | summ; done; } > "$OUT/placement_runs.txt"
{ echo "# the same with --placement-candidates 1 (plain alloc(in); alloc(out)), $NQ processes";
  for i in $(seq $NQ); do python3 bench.py --no-secondary --cpu-seconds 0 --parity-frames 0 --placement-candidates 1 2>/dev/null | tail -1 | summ; done; } > "$OUT/plain_alloc_runs.txt"
echo "== N=65536: the persistent launch against the two tiled launches (4096 packed frames, then config 3; Hann), and their crossover"
{ python3 tools/fused_probe.py 4096 65536 hann; python3 tools/fused_probe.py 18749 32768 hann; } > "$OUT/fused64k_vs_tiled.log" 2>&1
{ python3 experiments/fused64k_policy/crossover.py 32768; python3 experiments/fused64k_policy/crossover.py 65536; } > "$OUT/fused64k_crossover.log" 2>&1
cat "$OUT/fused64k_vs_tiled.log"; tail -3 "$OUT/placement_runs.txt"; tail -3 "$OUT/plain_alloc_runs.txt"
if [ "$LIGHT" != 1 ]; then
echo "== overlap of the two large-frame passes (SDRK_PLAN_OVERLAP_PASSES): sweep, then kernel traces"
timeout -k 10 400 python3 tools/overlap_probe.py both > "$OUT/overlap_probe.log" 2>&1; tail -4 "$OUT/overlap_probe.log"
( cd /tmp && export TMPDIR=/tmp
  for cfg in cfg3 cfg5; do for mode in serial overlap; do
    rocprofv3 --kernel-trace --output-format csv -d "$OUT/trace_${cfg}_$mode" -- python3 "$ROOT/tools/overlap_trace.py" $cfg $mode 2> /dev/null
    python3 "$ROOT/tools/summarise_overlap_trace.py" "$OUT/trace_${cfg}_$mode" "$cfg $mode"
  done; done ) > "$OUT/overlap_trace_summary.txt" 2>&1
cp "$OUT"/trace_cfg3_overlap/*/*kernel_trace.csv "$OUT/cfg3_overlap_kernel_trace.csv" 2>/dev/null
python3 - "$OUT/cfg3_overlap_kernel_trace.csv" <<'PY'
# keep the trace small enough to commit: the pass kernels of the last launch only, name shortened
import csv, sys
rows = [r for r in csv.DictReader(open(sys.argv[1])) if "_pass_" in r["Kernel_Name"]]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
rows = rows[-2 * 98:]
t0 = int(rows[0]["Start_Timestamp"])
with open(sys.argv[1], "w", newline="") as fh:
    w = csv.writer(fh)
    w.writerow(["kernel", "queue_id", "start_us", "end_us", "grid", "workgroup"])
    for r in rows:
        w.writerow(["col_pass" if "col_pass" in r["Kernel_Name"] else "row_pass", r.get("Queue_Id", ""),
                    round((int(r["Start_Timestamp"]) - t0) / 1e3, 2), round((int(r["End_Timestamp"]) - t0) / 1e3, 2),
                    r.get("Grid_Size", ""), r.get("Workgroup_Size", "")])
PY
rm -rf "$OUT"/trace_*/
cat "$OUT/overlap_trace_summary.txt"
echo "== host link: pageable / registered / pinned copies of 1 GiB in + 0.5 GiB out"
mkdir -p sdr-iq-visualizer_amd/build_tools
hipcc --offload-arch=gfx950 -O3 -o sdr-iq-visualizer_amd/build_tools/pcie_probe experiments/probes/pcie_probe.hip 2>/dev/null && ./sdr-iq-visualizer_amd/build_tools/pcie_probe > "$OUT/pcie_probe.log" 2>&1; cat "$OUT/pcie_probe.log"
fi
echo "== per-row measurements at the numpy boundary"; python3 tools/feat_host_probe.py > "$OUT/feat_host_probe.log" 2>&1; tail -2 "$OUT/feat_host_probe.log"
echo "== bench.py under torch.distributed.run, one rank (the nccl group formed and proven)"
python3 -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29641 bench.py --gpus 1 --steps 5 --warmup 1 --no-secondary --cpu-seconds 0 > "$OUT/bench_torchrun1_nccl.json" 2> "$OUT/bench_torchrun1_nccl.err"; echo "rc=$? $(wc -l < "$OUT/bench_torchrun1_nccl.json") line(s)"
echo "== bench.py --gpus 2, self-launched (rehearsal on one GPU: gloo rendezvous, ranks share the device)"
python3 bench.py --gpus 2 --steps 3 --warmup 1 --frames 262144 --cpu-seconds 3 --cpu-all-cores-seconds 2 2> "$OUT/bench_gpus2_rehearsal.err" | grep '^{' > "$OUT/bench_gpus2_rehearsal.json"; echo "rc=$? $(wc -c < "$OUT/bench_gpus2_rehearsal.json") bytes"
echo "== bench.py --gpus 4, self-launched on the one GPU (the control path of configs[3] / [4]; the -m gpu suite runs the same)"
python3 bench.py --gpus 4 --steps 3 --warmup 1 --frames 65536 --first-frame 3145728 --cpu-seconds 0 --placement-candidates 1 2> "$OUT/bench_gpus4_rehearsal.err" | grep '^{' > "$OUT/bench_gpus4_rehearsal.json"; echo "rc=$? $(wc -c < "$OUT/bench_gpus4_rehearsal.json") bytes"
echo "== bench.py --gpus 5, self-launched on the one GPU: the most ranks this pool's process guard admits beside the launcher (6 processes on"
echo "   the card; a 6-rank rehearsal was killed by it in round 6: profiles/r06/bench_gpus7_refused.txt).  frames 4096 per rank as the driver's"
echo "   8-GPU command line would be rehearsed; first frame 3 x 2^20 so that every rank's samples lie past index 2^32"
python3 bench.py --gpus 5 --frames 4096 --first-frame 3145728 --steps 3 --warmup 1 --cpu-seconds 0 --placement-candidates 1 2> "$OUT/bench_gpus5_rehearsal.err" | grep '^{' > "$OUT/bench_gpus5_rehearsal.json"; echo "rc=$? $(wc -c < "$OUT/bench_gpus5_rehearsal.json") bytes"
echo "== one-frame host call"; python3 tools/small_call_probe.py > "$OUT/small_call_probe.log" 2>&1; cat "$OUT/small_call_probe.log"
echo "== gpu tests"; python3 -m pytest tests -m gpu -q > "$OUT/pytest_gpu.log" 2>&1; tail -2 "$OUT/pytest_gpu.log"
