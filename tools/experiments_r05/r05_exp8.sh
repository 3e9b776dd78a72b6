#!/bin/bash
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r05_exp8
mkdir -p $OUT
cd $ROOT
timeout -k 10 500 python3 -m pytest tests -m gpu -q -x -k "features or peak or threshold or corner or classifier" > $OUT/pytest.out 2>&1; tail -3 $OUT/pytest.out | tee -a $OUT/log.txt
timeout -k 10 300 python3 tools/fused_vs_alone.py > $OUT/fva.out 2>&1; cat $OUT/fva.out | tee -a $OUT/log.txt
timeout -k 10 300 python3 tools/stress_features.py 900 811 > $OUT/sf.out 2>&1; tail -1 $OUT/sf.out | tee -a $OUT/log.txt
timeout -k 10 300 python3 tools/stress_fused.py 500 812 > $OUT/sfu.out 2>&1; tail -1 $OUT/sfu.out | cut -c1-200 | tee -a $OUT/log.txt
for r in 1 2; do for v in f1prev head; do
  lib=$ROOT/sdr-iq-visualizer_amd/lib_$v/libsdrk.so; [ $v = head ] && lib=$ROOT/sdr-iq-visualizer_amd/lib/libsdrk.so
  SDRK_LIB=$lib timeout -k 10 200 python3 tools/feat_probe.py > $OUT/${v}_$r.out 2>&1
  echo "$v: $(grep warm $OUT/${v}_$r.out | sed 's/fused //' | tr '\n' ' ')" | tee -a $OUT/log.txt
done; done
echo done | tee -a $OUT/log.txt
