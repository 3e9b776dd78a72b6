#!/bin/bash
# gpurun call 6: bitmask knock-outs of the timing-only build (ALIAS ring, NOWAIT, ring depth 1), 4096 packed frames.
#   bits: 1 all col, 2 all row, 4 no ring traffic, 8 no transforms, 16 no input loads, 32 no output stores
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r06_fused6; mkdir -p $OUT
cd $ROOT
for k in 0 4 8 12 16 32 48 60 17 18 9 10 34; do
  export SDRK_LIB=$ROOT/sdr-iq-visualizer_amd/lib_fuexp_alias_k${k}_d1/libsdrk.so
  timeout -k 10 200 python3 experiments/fused64k_policy/sweep.py 4096 65536 --nowait-only --few > $OUT/knock_k$k.log 2>&1 || { echo k$k failed; tail -5 $OUT/knock_k$k.log; exit 1; }
  echo "== knock-out mask $k: $(grep '^  3       2      2' $OUT/knock_k$k.log)  $(grep '^  1       2      2' $OUT/knock_k$k.log)"
done
