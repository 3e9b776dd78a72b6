#!/bin/bash
# Developer helper (GPU box): mean FETCH_SIZE / WRITE_SIZE per dispatch of the tiled-pass kernels of one config.
#   tools/kpmc.sh "1048576 256 1048576 hann" base v1 ...
# Read bytes = 2 x FETCH_SIZE KB x 1024 on gfx950 (the counter tallies a 128-byte request as 64); WRITE_SIZE KB x 1024.
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
CFG=$1; shift
cd /tmp && export TMPDIR=/tmp
for v in "$@"; do
  lib=$ROOT/sdr-iq-visualizer_amd/lib_$v/libsdrk.so; [ "$v" = base ] && lib=$ROOT/sdr-iq-visualizer_amd/lib/libsdrk.so
  for c in FETCH_SIZE WRITE_SIZE; do
    rm -rf /tmp/kp_${v}_$c
    SDRK_LIB=$lib rocprofv3 --pmc $c --output-format csv -d /tmp/kp_${v}_$c -- python3 $ROOT/tools/one_config.py $CFG > /dev/null 2>&1
    python3 - "$v" "$c" <<PY
import csv,glob,sys,collections
v,c=sys.argv[1:3]
acc=collections.defaultdict(list)
for f in glob.glob("/tmp/kp_%s_%s/*/*counter_collection.csv" % (v,c)):
    per=collections.defaultdict(float)
    for r in csv.DictReader(open(f)):
        if r["Counter_Name"]==c and "_pass_" in r["Kernel_Name"]:
            per[(r["Dispatch_Id"],r["Kernel_Name"])]+=float(r["Counter_Value"])
    for (d,k),x in per.items(): acc[k].append(x)
for k,xs in acc.items():
    kb=sum(xs)/len(xs); mb=kb*1024*(2 if c=="FETCH_SIZE" else 1)/1e6
    print(v, c, k[6:48], len(xs), "mean_MB", round(mb,2))
PY
  done
done
