#!/bin/bash
# round 5, experiment 2 (GPU box): warm A/B of the col-pass variants, warm chunk sweep, counters before / after
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r05_exp2
mkdir -p $OUT
cd $ROOT
step() {
    local name=$1 to=$2; shift 2
    echo "== $name" | tee -a $OUT/log.txt
    timeout -k 10 $to "$@" > $OUT/$name.out 2> $OUT/$name.err
    local rc=$?
    echo "rc=$rc" | tee -a $OUT/log.txt
    if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "timeout in $name: stopping" | tee -a $OUT/log.txt; exit 1; fi
    return $rc
}
step ab_warm 600 python3 tools/ab_steady.py --rounds 3 base e1 e2; cat $OUT/ab_warm.out | tee -a $OUT/log.txt
step chunk_warm 600 python3 tools/ab_steady.py --rounds 2 base:SDRK_SCRATCH_MB=128 base:SDRK_SCRATCH_MB=160 base:SDRK_SCRATCH_MB=192 base:SDRK_SCRATCH_MB=224 base:SDRK_SCRATCH_MB=256; cat $OUT/chunk_warm.out | tee -a $OUT/log.txt
step cfg3_warm 300 python3 tools/ab_steady.py --rounds 2 --cfg "65536 18749 32768 hann" --transforms 10 base; cat $OUT/cfg3_warm.out | tee -a $OUT/log.txt
cd /tmp && export TMPDIR=/tmp
CFG="1048576 256 1048576 hann --transforms 6 --warm-ms 60"
for v in base e1; do
  lib=$ROOT/sdr-iq-visualizer_amd/lib_$v/libsdrk.so; [ $v = base ] && lib=$ROOT/sdr-iq-visualizer_amd/lib/libsdrk.so
  for c in GRBM_GUI_ACTIVE TCC_EA0_WRREQ_STALL_sum TCC_EA0_RDREQ_DRAM_CREDIT_STALL_sum TCC_BUSY_sum TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum TCC_EA0_RDREQ_LEVEL_sum TCP_PENDING_STALL_CYCLES_sum TCC_HIT_sum TCC_MISS_sum TCC_TAG_STALL_sum; do
    SDRK_LIB=$lib step pmc_${v}_$c 200 rocprofv3 --pmc $c --output-format csv -d $OUT/pmc_$v/$c -- python3 $ROOT/tools/cfg_steady.py $CFG || break
  done
  SDRK_LIB=$lib step sq_${v} 200 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM --output-format csv -d $OUT/pmc_$v/SQ -- python3 $ROOT/tools/cfg_steady.py $CFG
done
cd $ROOT
python3 - $OUT <<'PY' | tee -a $OUT/log.txt
import csv, glob, sys, collections, re
out = sys.argv[1]
for v in ("base", "e1"):
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    for f in glob.glob(f"{out}/pmc_{v}/*/*/*counter_collection.csv"):
        per = collections.defaultdict(float)
        for r in csv.DictReader(open(f)):
            m = re.search(r"sdrk::(\w+)", r["Kernel_Name"])
            if m:
                per[(r["Dispatch_Id"], m.group(1), r["Counter_Name"])] += float(r["Counter_Value"])
        for (d, k, c), x in per.items():
            acc[k][c].append(x)
    print("variant", v)
    for k, cs in sorted(acc.items()):
        print(" ", k)
        for c, vs in sorted(cs.items()):
            vs = sorted(vs)
            print("     %-40s n=%-4d median=%-14.6g" % (c, len(vs), vs[len(vs) // 2]))
PY
echo done | tee -a $OUT/log.txt
