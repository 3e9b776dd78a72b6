#!/bin/bash
# gpurun call 1 of the round-6 fused-kernel experiment: policy sweep (ring depth 2, then 4) on 4096 packed frames, then counters.
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r06_fused1; mkdir -p $OUT
cd $ROOT
export SDRK_LIB=$ROOT/sdr-iq-visualizer_amd/lib_fuexp/libsdrk.so
timeout -k 10 420 python3 experiments/fused64k_policy/sweep.py 4096 65536 > $OUT/sweep_d2_packed.log 2>&1 || { echo sweep failed; tail -5 $OUT/sweep_d2_packed.log; exit 1; }
tail -64 $OUT/sweep_d2_packed.log
bash experiments/fused64k_policy/pmc.sh $OUT/pmc_d2 4096 65536 tiled 3:2:2:0 2:2:2:0 1:2:2:0 1:0:0:0 1:2:17:0 1:18:19:0 3:2:17:0 3:0:0:0 1:2:2:1 3:2:2:1 > $OUT/pmc_d2_packed.log 2>&1 || { echo pmc failed; tail $OUT/pmc_d2_packed.log; exit 1; }
cat $OUT/pmc_d2_packed.log
export SDRK_LIB=$ROOT/sdr-iq-visualizer_amd/lib_fuexp_d4/libsdrk.so
timeout -k 10 300 python3 experiments/fused64k_policy/sweep.py 4096 65536 --quick > $OUT/sweep_d4_packed_quick.log 2>&1 || { echo sweep d4 failed; tail -5 $OUT/sweep_d4_packed_quick.log; exit 1; }
tail -14 $OUT/sweep_d4_packed_quick.log
