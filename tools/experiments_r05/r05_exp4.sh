#!/bin/bash
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r05_exp4
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
step fused_vs_alone 300 python3 tools/fused_vs_alone.py; cat $OUT/fused_vs_alone.out | tee -a $OUT/log.txt
step pytest_gpu 1100 python3 -m pytest tests -m gpu -q --durations=8; tail -25 $OUT/pytest_gpu.out | tee -a $OUT/log.txt
echo done | tee -a $OUT/log.txt
