#!/bin/bash
# SQ counters of the flagship, packed complex arithmetic (what ships) against the scalar build: instructions issued and
# where the waves' cycles go (separate --pmc passes, kernel trace off: the pool's rule).  One short bench process per pass.
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r05_exp13
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
ARGS="--steps 3 --warmup 1 --cpu-seconds 0 --parity-frames 0 --no-secondary --placement-candidates 1"
for v in packed scalar; do
  lib=$ROOT/sdr-iq-visualizer_amd/lib/libsdrk.so; [ $v = scalar ] && lib=$ROOT/sdr-iq-visualizer_amd/lib_scalar/libsdrk.so
  export SDRK_LIB=$lib
  rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU --output-format csv -d $OUT/${v}_sq1 -- python3 $ROOT/bench.py $ARGS > /dev/null 2> $OUT/${v}_sq1.err
  rocprofv3 --pmc SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM SQ_INSTS_SMEM --output-format csv -d $OUT/${v}_sq2 -- python3 $ROOT/bench.py $ARGS > /dev/null 2> $OUT/${v}_sq2.err
  rocprofv3 --pmc SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_ANY --output-format csv -d $OUT/${v}_sq3 -- python3 $ROOT/bench.py $ARGS > /dev/null 2> $OUT/${v}_sq3.err
  rocprofv3 --pmc SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_WAIT_ANY SQ_LDS_BANK_CONFLICT --output-format csv -d $OUT/${v}_sq4 -- python3 $ROOT/bench.py $ARGS > /dev/null 2> $OUT/${v}_sq4.err
done
unset SDRK_LIB
python3 - $OUT <<'PY' | tee $OUT/summary.txt
import csv, glob, os, sys
from collections import defaultdict
root = sys.argv[1]
for v in ("packed", "scalar"):
    acc = defaultdict(list)
    for d in sorted(glob.glob(os.path.join(root, v + "_sq*"))):
        if not os.path.isdir(d): continue
        for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
            per = defaultdict(float)
            for row in csv.DictReader(open(f)):
                if "fft4096_kernel" in row["Kernel_Name"]:
                    per[(row["Dispatch_Id"], row["Counter_Name"])] += float(row["Counter_Value"])
            for (_, c), val in per.items():
                acc[c].append(val)
    m = {c: sum(x) / len(x) for c, x in acc.items()}
    waves = m.get("SQ_WAVES", 0)
    frames_per_wave = (1 << 20) / (waves / 4) if waves else float("nan")      # 4 waves per workgroup, one frame per workgroup pass
    print(f"== {v}: dispatches {max(len(x) for x in acc.values()) if acc else 0}, waves per dispatch {waves:.0f} (each workgroup transforms {frames_per_wave:.1f} frames)")
    for c in sorted(m):
        if c == "SQ_WAVES": continue
        per_wave_frame = m[c] / waves / frames_per_wave if waves else float("nan")
        print(f"   {c:24s} {m[c]:16.0f} per dispatch   {per_wave_frame:10.1f} per wave and frame")
PY
