#!/bin/bash
# Developer helper (GPU box): per-kernel average times of one config under rocprofv3 for several library variants.
#   tools/ktimes.sh "1048576 256 1048576 hann" base v1 v2 ...
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
CFG=$1; shift
cd /tmp && export TMPDIR=/tmp
for v in "$@"; do
  lib=$ROOT/sdr-iq-visualizer_amd/lib_$v/libsdrk.so; [ "$v" = base ] && lib=$ROOT/sdr-iq-visualizer_amd/lib/libsdrk.so
  rm -rf /tmp/kt_$v
  SDRK_LIB=$lib rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/kt_$v -- python3 $ROOT/tools/one_config.py $CFG > /dev/null 2>&1
  python3 - "$v" <<PY
import csv,glob,sys
for f in glob.glob("/tmp/kt_%s/*/*kernel_stats.csv" % sys.argv[1]):
    for r in csv.DictReader(open(f)):
        if "_pass_" in r["Name"]:
            print(sys.argv[1], r["Name"][11:50], r["Calls"], "avg_us", round(float(r["AverageNs"])/1e3,2))
PY
done
