#!/bin/bash
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r05_exp5
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
step pytest_feat 600 python3 -m pytest tests -m gpu -q -x -k "features or peak or threshold or corner or classifier"; tail -3 $OUT/pytest_feat.out | tee -a $OUT/log.txt
step fused_vs_alone 300 python3 tools/fused_vs_alone.py; cat $OUT/fused_vs_alone.out | tee -a $OUT/log.txt
for r in 1 2; do for v in f1r04 f1base head; do
  lib=$ROOT/sdr-iq-visualizer_amd/lib_$v/libsdrk.so; [ $v = head ] && lib=$ROOT/sdr-iq-visualizer_amd/lib/libsdrk.so
  SDRK_LIB=$lib step feat_${v}_$r 200 python3 tools/feat_probe.py; echo "$v: $(grep warm $OUT/feat_${v}_$r.out | tr '\n' ' ')" | tee -a $OUT/log.txt
done; done
step stress_features 400 python3 tools/stress_features.py; tail -3 $OUT/stress_features.out | tee -a $OUT/log.txt
echo done | tee -a $OUT/log.txt
