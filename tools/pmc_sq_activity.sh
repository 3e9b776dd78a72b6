#!/bin/bash
# Developer helper (GPU box): SQ activity counters of one configuration's kernels, one rocprofv3 --pmc pass per set.
#   tools/pmc_sq_activity.sh "65536 18749 32768 hann" outdir
# (the memory-pipeline counters — TCC / TCP / TA — are tools/pmc_mem_counters.sh's, ONE per pass)
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
CFG=$1; OUT=${2:-$ROOT/gpurun_out/pmc_mem}
case "$OUT" in /*) ;; *) OUT="$ROOT/$OUT";; esac
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
i=0
# (round 3 also put several TA_* / TCP_* / TCC_* "_sum" counters into one set here; rocprofiler refused the SET —
#  "error code 38: Request exceeds the capabilities of the hardware to collect", the TCC block has 4 slots per pass —
#  and aborted the process.  The counters themselves collect fine one per pass: tools/pmc_mem_counters.sh.)
# (the SQ block has 8 slots per pass: MI355X_MICROARCH.md "rocprofv3 PMC slots"; WAIT_ANY + WAIT_INST_ANY + ACTIVE_INST_ANY ~ WAVE_CYCLES)
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM" \
           "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS"; do
  i=$((i+1))
  timeout -k 10 240 rocprofv3 --pmc $set --output-format csv -d "$OUT/set$i" -- python3 "$ROOT/tools/one_config.py" $CFG > /dev/null 2> "$OUT/set$i.err" || { echo "set $i failed: $set"; tail -3 "$OUT/set$i.err"; break; }
done
python3 - "$OUT" <<'PY'
import csv,glob,sys,collections,re
out=sys.argv[1]
acc=collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(out+"/set*/*/*counter_collection.csv"):
    per=collections.defaultdict(float)
    for r in csv.DictReader(open(f)):
        m=re.search(r"sdrk::(\w+)", r["Kernel_Name"])
        if not m: continue
        per[(r["Dispatch_Id"],m.group(1),r["Counter_Name"])]+=float(r["Counter_Value"])
    for (d,k,c),v in per.items(): acc[k][c].append(v)
for k,cs in acc.items():
    print(k)
    for c,vs in sorted(cs.items()): print("   %-40s n=%d mean=%.5g" % (c,len(vs),sum(vs)/len(vs)))
PY
