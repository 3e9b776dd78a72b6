#!/bin/bash
# gpurun call 4: the hand-over's flags kept in the L2 (plain stores instead of agent-scope ones), ring depth 2 and 1; counters of
# the depth-1 rings (3 sets x 512 KiB = 1.5 MiB, 2 sets = 1 MiB per XCD).
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r06_fused4; mkdir -p $OUT
cd $ROOT
for v in flagl2 flagl2_d1 few_d1; do
  export SDRK_LIB=$ROOT/sdr-iq-visualizer_amd/lib_fuexp_$v/libsdrk.so
  timeout -k 10 200 python3 experiments/fused64k_policy/sweep.py 4096 65536 --few > $OUT/sweep_$v.log 2>&1 || { echo $v failed; tail -5 $OUT/sweep_$v.log; exit 1; }
  echo "== $v"; tail -8 $OUT/sweep_$v.log
done
export SDRK_LIB=$ROOT/sdr-iq-visualizer_amd/lib_fuexp_flagl2_d1/libsdrk.so
bash experiments/fused64k_policy/pmc.sh $OUT/pmc_flagl2_d1 4096 65536 3:2:17:0 2:2:17:0 3:2:2:0 > $OUT/pmc_flagl2_d1.log 2>&1 || { echo pmc failed; tail $OUT/pmc_flagl2_d1.log; exit 1; }
cat $OUT/pmc_flagl2_d1.log
