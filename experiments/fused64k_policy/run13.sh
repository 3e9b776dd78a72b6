#!/bin/bash
# gpurun call 13: role placement — col and row workgroups mixed on every CU (FU_ROLE_MIX) against the ticket order's natural split
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r06_fused13; mkdir -p $OUT
cd $ROOT
for v in few mix few mix; do
  export SDRK_LIB=$ROOT/sdr-iq-visualizer_amd/lib_fuexp_$v/libsdrk.so
  timeout -k 10 200 python3 experiments/fused64k_policy/sweep.py 18749 32768 --few > $OUT/sweep_cfg3_$v.log 2>&1 || { echo $v failed; tail -5 $OUT/sweep_cfg3_$v.log; exit 1; }
  echo "== $v cfg3: $(grep '^  3       2      2' $OUT/sweep_cfg3_$v.log)"
  timeout -k 10 200 python3 experiments/fused64k_policy/sweep.py 4096 65536 --few > $OUT/sweep_packed_$v.log 2>&1 || { echo $v failed; exit 1; }
  echo "== $v packed: $(grep '^  3       2      2' $OUT/sweep_packed_$v.log)"
done
