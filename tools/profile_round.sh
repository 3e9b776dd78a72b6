#!/bin/bash
# Collect the rocprofv3 evidence of one round on the GPU box (run through gpurun from the repo root):
#   tools/profile_round.sh r02
# kernel-trace + stats of the default bench, FETCH_SIZE / WRITE_SIZE PMC passes (separate runs, as the
# MI355X guide prescribes) of the bench and of BASELINE configs 3 and 5.  Outputs under gpurun_out/prof_<tag>/;
# tools/summarise_profiles.py turns them into the files committed under profiles/<tag>/.
set -u
TAG=${1:-r06}
LIGHT=${SDRK_COLLECT_LIGHT:-0}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/prof_$TAG
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
run() {  # name, rocprof args..., -- program args
    local name=$1; shift
    echo "== $name" >&2
    rocprofv3 "$@" > "$OUT/$name.stdout" 2> "$OUT/$name.stderr" || echo "rocprofv3 $name failed rc=$?" >&2
}
BENCH_SHORT="--steps 3 --warmup 1 --cpu-seconds 0 --parity-frames 0 --no-secondary"
run bench_trace --kernel-trace --stats --output-format csv -d "$OUT/bench_trace" -- python3 "$ROOT/bench.py" --no-secondary --cpu-seconds 0
run bench_fetch --pmc FETCH_SIZE --output-format csv -d "$OUT/bench_fetch" -- python3 "$ROOT/bench.py" $BENCH_SHORT
run bench_write --pmc WRITE_SIZE --output-format csv -d "$OUT/bench_write" -- python3 "$ROOT/bench.py" $BENCH_SHORT
# BASELINE configs 3 and 5, and the frame lengths DESIGN.md names as the slowest passes left (not BASELINE configs)
CFGS=("cfg3 65536 18749 32768 hann" "cfg5 1048576 256 1048576 hann" "n16384 16384 8192 16384 rect" "n2p21 2097152 128 2097152 rect" "n2p22 4194304 64 4194304 rect")
[ "$LIGHT" = 1 ] && CFGS=("cfg3 65536 18749 32768 hann" "cfg5 1048576 256 1048576 hann")   # the BASELINE configs only
for cfg in "${CFGS[@]}"; do
    set -- $cfg
    # (round 5: tools/cfg_steady.py instead of one_config.py — warmed up by time, transforms back to back as bench.py times
    #  them; one_config.py's one warm-up + three isolated launches ran in the clock ramp after idle and gave kernel times
    #  11-15 % above the event-timed step they were meant to explain)
    run $1_trace --kernel-trace --stats --output-format csv -d "$OUT/$1_trace" -- python3 "$ROOT/tools/cfg_steady.py" $2 $3 $4 $5 --transforms 12 --out "$OUT/$1_steady.json"
    python3 "$ROOT/tools/summarise_cfg_trace.py" "$OUT/$1_trace" "$OUT/$1_steady.json" > "$OUT/$1_trace_vs_events.json" 2>> "$OUT/summary.err"
    run $1_fetch --pmc FETCH_SIZE --output-format csv -d "$OUT/$1_fetch" -- python3 "$ROOT/tools/cfg_steady.py" $2 $3 $4 $5 --transforms 4 --warm-ms 40
    run $1_write --pmc WRITE_SIZE --output-format csv -d "$OUT/$1_write" -- python3 "$ROOT/tools/cfg_steady.py" $2 $3 $4 $5 --transforms 4 --warm-ms 40
done
# BASELINE config 3 in its other form, the two tiled launches (the default plan above takes the one persistent launch since round 6)
run cfg3tiled_trace --kernel-trace --stats --output-format csv -d "$OUT/cfg3tiled_trace" -- python3 "$ROOT/tools/cfg_steady.py" 65536 18749 32768 hann --form tiled --transforms 12 --out "$OUT/cfg3tiled_steady.json"
python3 "$ROOT/tools/summarise_cfg_trace.py" "$OUT/cfg3tiled_trace" "$OUT/cfg3tiled_steady.json" > "$OUT/cfg3tiled_trace_vs_events.json" 2>> "$OUT/summary.err"
run cfg3tiled_fetch --pmc FETCH_SIZE --output-format csv -d "$OUT/cfg3tiled_fetch" -- python3 "$ROOT/tools/cfg_steady.py" 65536 18749 32768 hann --form tiled --transforms 4 --warm-ms 40
run cfg3tiled_write --pmc WRITE_SIZE --output-format csv -d "$OUT/cfg3tiled_write" -- python3 "$ROOT/tools/cfg_steady.py" 65536 18749 32768 hann --form tiled --transforms 4 --warm-ms 40
# the per-row reductions: kernel trace + SQ instruction / activity counters, separate passes
run feat_trace --kernel-trace --stats --output-format csv -d "$OUT/feat_trace" -- python3 "$ROOT/tools/feat_probe.py"
run feat_sq1 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU --output-format csv -d "$OUT/feat_sq1" -- python3 "$ROOT/tools/feat_probe.py"
run feat_sq2 --pmc SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM SQ_INSTS_SMEM --output-format csv -d "$OUT/feat_sq2" -- python3 "$ROOT/tools/feat_probe.py"
run feat_sq3 --pmc SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_ANY --output-format csv -d "$OUT/feat_sq3" -- python3 "$ROOT/tools/feat_probe.py"
run feat_sq4 --pmc SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_WAIT_ANY SQ_LDS_BANK_CONFLICT --output-format csv -d "$OUT/feat_sq4" -- python3 "$ROOT/tools/feat_probe.py"
python3 "$ROOT/tools/summarise_profiles.py" "$OUT" > "$OUT/summary.json" 2> "$OUT/summary.err"
python3 "$ROOT/tools/summarise_profiles.py" "$OUT" --rows "$OUT/pmc_fetch_write_rows.csv" 2>> "$OUT/summary.err"
cat "$OUT/summary.json"
