#!/bin/bash
# gpurun call 3: knock-outs of the timing-only build (ALIAS ring + NOWAIT): where do its 1.08 ms go, if not into the fabric?
#   k1 every workgroup a col workgroup   k2 every workgroup a row workgroup   k3 no ring traffic   k4 no transforms
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r06_fused3; mkdir -p $OUT
cd $ROOT
for k in 1 2 3 4; do
  export SDRK_LIB=$ROOT/sdr-iq-visualizer_amd/lib_fuexp_alias_k$k/libsdrk.so
  timeout -k 10 200 python3 experiments/fused64k_policy/sweep.py 4096 65536 --nowait-only --few > $OUT/knock_k$k.log 2>&1 || { echo k$k failed; tail -5 $OUT/knock_k$k.log; exit 1; }
  echo "== knock-out $k"; tail -8 $OUT/knock_k$k.log
done
