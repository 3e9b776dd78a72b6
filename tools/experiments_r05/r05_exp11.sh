#!/bin/bash
# long rows (the generic peak scan, spacings > 16): candidate reads without short-circuit branches — tests, then the stand-alone kernel per row length
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r05_exp11
mkdir -p $OUT
cd $ROOT
timeout -k 10 500 python3 -m pytest tests -m gpu -q -x -k "features or peak or threshold or corner or classifier" > $OUT/pytest.out 2>&1; tail -2 $OUT/pytest.out | tee -a $OUT/log.txt
timeout -k 10 300 python3 tools/stress_features.py 900 911 > $OUT/sf.out 2>&1; tail -1 $OUT/sf.out | cut -c1-160 | tee -a $OUT/log.txt
for r in 1 2; do for v in rowprev head; do
  lib=$ROOT/sdr-iq-visualizer_amd/lib_$v/libsdrk.so; [ $v = head ] && lib=$ROOT/sdr-iq-visualizer_amd/lib/libsdrk.so
  echo "== $v" | tee -a $OUT/log.txt
  SDRK_LIB=$lib timeout -k 10 300 python3 tools/feat_rows_probe.py 2>&1 | tee -a $OUT/log.txt
done; done
echo done | tee -a $OUT/log.txt
