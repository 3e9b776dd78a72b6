#!/bin/bash
# gpurun call 5: plain-flag hand-over on the config-3 shape (18749 frames, hop 32768), ring depth 1 and 2; counters at depth 1.
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r06_fused5; mkdir -p $OUT
cd $ROOT
for v in flagl2_d1 flagl2; do
  export SDRK_LIB=$ROOT/sdr-iq-visualizer_amd/lib_fuexp_$v/libsdrk.so
  timeout -k 10 300 python3 experiments/fused64k_policy/sweep.py 18749 32768 --few > $OUT/sweep_cfg3_$v.log 2>&1 || { echo $v failed; tail -5 $OUT/sweep_cfg3_$v.log; exit 1; }
  echo "== $v"; tail -8 $OUT/sweep_cfg3_$v.log
done
export SDRK_LIB=$ROOT/sdr-iq-visualizer_amd/lib_fuexp_flagl2_d1/libsdrk.so
bash experiments/fused64k_policy/pmc.sh $OUT/pmc_cfg3_flagl2_d1 18749 32768 3:2:2:0 3:2:17:0 2:2:17:0 > $OUT/pmc_cfg3_flagl2_d1.log 2>&1 || { echo pmc failed; tail $OUT/pmc_cfg3_flagl2_d1.log; exit 1; }
cat $OUT/pmc_cfg3_flagl2_d1.log
