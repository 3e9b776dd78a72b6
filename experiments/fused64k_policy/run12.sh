#!/bin/bash
# gpurun call 12: s_sleep between polls that made no progress (0 / 1 / 2 / 4 / 8), product hand-over, config-3 shape and packed.
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r06_fused12; mkdir -p $OUT
cd $ROOT
for v in few sl0 sl2 sl4 sl8 few; do
  export SDRK_LIB=$ROOT/sdr-iq-visualizer_amd/lib_fuexp_$v/libsdrk.so
  timeout -k 10 200 python3 experiments/fused64k_policy/sweep.py 18749 32768 --few > $OUT/sweep_cfg3_$v.log 2>&1 || { echo $v failed; tail -5 $OUT/sweep_cfg3_$v.log; exit 1; }
  echo "== $v cfg3: $(grep '^  3       2      2' $OUT/sweep_cfg3_$v.log)"
  timeout -k 10 200 python3 experiments/fused64k_policy/sweep.py 4096 65536 --few > $OUT/sweep_packed_$v.log 2>&1 || { echo $v failed; exit 1; }
  echo "== $v packed: $(grep '^  3       2      2' $OUT/sweep_packed_$v.log)"
done
