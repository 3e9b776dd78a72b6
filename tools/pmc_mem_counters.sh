#!/bin/bash
# Developer helper (GPU box): memory-pipeline counters (L2 = TCC, vector L1 = TCP, address unit = TA) of one
# configuration's kernels — ONE counter per rocprofv3 --pmc pass.
#
#   tools/pmc_mem_counters.sh "65536 18749 32768 hann" gpurun_out/pmc_mem_cfg3
#
# Why one per pass: the TCC block of gfx950 has 4 counter slots per pass (MI355X_MICROARCH.md, "rocprofv3 PMC slots")
# and a derived `_sum` counter takes one slot in every instance; round 3 asked for several TCC / TCP / TA counters in
# one set and rocprofiler refused the set (`rocprofiler_create_counter_config … error code 38: Request exceeds the
# capabilities of the hardware to collect`, gpurun_out/pmc_cfg3/set3.err) and aborted the process.  The counters
# themselves are fine — FETCH_SIZE / WRITE_SIZE are TCC-derived too and have always been collected one per pass.
#
# The program comes directly after `--`; every pass is bounded by its own timeout; the first pass that fails ends the
# script (no further GPU step after a failure).
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
CFG=$1; OUT=${2:-$ROOT/gpurun_out/pmc_mem}
case "$OUT" in /*) ;; *) OUT="$ROOT/$OUT";; esac
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
COUNTERS=${SDRK_PMC_COUNTERS:-"
TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_READ_sum TCC_WRITE_sum TCC_STREAMING_REQ_sum
TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_DRAM_sum TCC_EA0_RDREQ_LEVEL_sum TCC_EA0_RDREQ_DRAM_CREDIT_STALL_sum
TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_DRAM_sum TCC_EA0_WRREQ_LEVEL_sum TCC_EA0_WRREQ_STALL_sum
TCC_EA0_WRREQ_DRAM_CREDIT_STALL_sum TCC_TOO_MANY_EA_WRREQS_STALL_sum TCC_TAG_STALL_sum TCC_BUSY_sum
TCC_NORMAL_WRITEBACK_sum
TCP_PENDING_STALL_CYCLES_sum TCP_TCC_READ_REQ_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_TCC_WRITE_REQ_sum
TCP_TCC_WRITE_REQ_LATENCY_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum TCP_READ_TAGCONFLICT_STALL_CYCLES_sum
TCP_GATE_EN1_sum TCP_TCP_LATENCY_sum
TA_TA_BUSY_sum TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum
GRBM_GUI_ACTIVE
"}
ok=0
for c in $COUNTERS; do
  echo "[pmc] $c"
  if ! timeout -k 10 240 rocprofv3 --pmc $c --output-format csv -d "$OUT/$c" -- python3 "$ROOT/tools/one_config.py" $CFG \
        > "$OUT/$c.out" 2> "$OUT/$c.err"; then
    echo "[pmc] pass failed: $c (see $OUT/$c.err) — stopping here"; tail -5 "$OUT/$c.err"
    break
  fi
  ok=$((ok+1))
done
echo "[pmc] $ok passes collected"
python3 - "$OUT" "$CFG" <<'PY'
import csv, glob, sys, collections, re, os
out, cfg = sys.argv[1], sys.argv[2]
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(out + "/*/*/*counter_collection.csv"):
    per = collections.defaultdict(float)
    for r in csv.DictReader(open(f)):
        m = re.search(r"sdrk::(\w+)", r["Kernel_Name"])
        if not m:
            continue
        per[(r["Dispatch_Id"], m.group(1), r["Counter_Name"])] += float(r["Counter_Value"])
    for (d, k, c), v in per.items():
        acc[k][c].append(v)
print("# one_config.py %s — mean per dispatch, one counter per rocprofv3 --pmc pass" % cfg)
for k, cs in sorted(acc.items()):
    print(k)
    for c, vs in sorted(cs.items()):
        vs = sorted(vs)
        print("   %-44s n=%-4d mean=%-12.6g median=%-12.6g" % (c, len(vs), sum(vs) / len(vs), vs[len(vs) // 2]))
PY
