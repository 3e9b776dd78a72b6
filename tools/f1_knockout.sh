#!/bin/bash
# Developer helper (GPU box): what each phase of the fused N = 4096 reductions costs — timing-only builds with one phase knocked out
# (-DSDRK_F1_SKIP=bit, tools/variant.sh ko<bit>; results of those builds are WRONG, only tools/feat_probe.py's times are read).
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/f1_knockout
mkdir -p $OUT
cd $ROOT
for r in 1 2; do for v in base ko2 ko4 ko8 ko32 ko64 ko96; do
  lib=$ROOT/sdr-iq-visualizer_amd/lib_$v/libsdrk.so; [ $v = base ] && lib=$ROOT/sdr-iq-visualizer_amd/lib/libsdrk.so
  SDRK_LIB=$lib timeout -k 10 200 python3 tools/feat_probe.py > $OUT/${v}_$r.out 2> $OUT/${v}_$r.err
  echo "$v: $(grep warm $OUT/${v}_$r.out | sed 's/fused //; s/ ms,.*ns\/row//' | tr '\n' ' ')" | tee -a $OUT/log.txt
done; done
echo done | tee -a $OUT/log.txt
