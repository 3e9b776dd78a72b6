#!/bin/bash
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r05_exp6
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
step ab5 400 python3 tools/ab_steady.py --rounds 3 base nofence; cat $OUT/ab5.out | tee -a $OUT/log.txt
step ab3 400 python3 tools/ab_steady.py --rounds 3 --cfg "65536 18749 32768 hann" --transforms 10 base nofence; cat $OUT/ab3.out | tee -a $OUT/log.txt
step ab19 400 python3 tools/ab_steady.py --rounds 2 --cfg "524288 512 524288 rect" --transforms 10 base nofence; cat $OUT/ab19.out | tee -a $OUT/log.txt
step sweep 900 python3 tools/size_sweep.py; cat $OUT/sweep.out | tee -a $OUT/log.txt
echo done | tee -a $OUT/log.txt
