#!/bin/bash
# round 3, GPU call 3: counters of the fused feature kernel
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r03c
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
rocprofv3 -L > "$OUT/counters_avail.txt" 2>&1
grep -c . "$OUT/counters_avail.txt"
for set in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU" "SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM SQ_INSTS_SMEM" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_ANY" "SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_WAIT_ANY SQ_LDS_BANK_CONFLICT" "SQ_INST_CYCLES_VMEM SQ_INSTS_VALU_MFMA_MOPS_F64 SQ_LDS_IDX_ACTIVE SQ_LDS_ATOMIC_RETURN"; do
  tag=$(echo $set | tr ' ' '_' | cut -c1-40)
  rocprofv3 --pmc $set --output-format csv -d "$OUT/pmc_$tag" -- python3 "$ROOT/tools/feat_probe.py" > "$OUT/pmc_$tag.stdout" 2> "$OUT/pmc_$tag.stderr" || echo "pmc $set failed"
done
python3 - "$OUT" <<'PY'
import csv,glob,sys,collections
out=sys.argv[1]
acc=collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(out+"/pmc_*/*/*counter_collection.csv"):
    per=collections.defaultdict(float)
    for r in csv.DictReader(open(f)):
        k=r["Kernel_Name"][:40]
        per[(r["Dispatch_Id"],k,r["Counter_Name"])]+=float(r["Counter_Value"])
    for (d,k,c),v in per.items(): acc[k][c].append(v)
for k,cs in acc.items():
    print(k)
    for c,vs in sorted(cs.items()): print("   %-32s n=%d mean=%.4g" % (c,len(vs),sum(vs)/len(vs)))
PY
