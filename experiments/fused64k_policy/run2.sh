#!/bin/bash
# gpurun call 2: (a) timing-only ALIAS build — all three sets of an XCD share one 1 MiB ring (no-wait; wrong results by
# construction): what full occupancy over an L2-resident ring would cost; its counters; (b) the config-3 shape (50 % overlap).
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r06_fused2; mkdir -p $OUT
cd $ROOT
export SDRK_LIB=$ROOT/sdr-iq-visualizer_amd/lib_fuexp_alias/libsdrk.so
timeout -k 10 300 python3 experiments/fused64k_policy/sweep.py 4096 65536 --nowait-only > $OUT/sweep_alias_packed.log 2>&1 || { echo sweep failed; tail -5 $OUT/sweep_alias_packed.log; exit 1; }
tail -62 $OUT/sweep_alias_packed.log
bash experiments/fused64k_policy/pmc.sh $OUT/pmc_alias 4096 65536 3:2:2:1 3:2:17:1 3:0:0:1 3:18:19:1 2:2:17:1 > $OUT/pmc_alias_packed.log 2>&1 || { echo pmc failed; tail $OUT/pmc_alias_packed.log; exit 1; }
cat $OUT/pmc_alias_packed.log
export SDRK_LIB=$ROOT/sdr-iq-visualizer_amd/lib_fuexp/libsdrk.so
timeout -k 10 400 python3 experiments/fused64k_policy/sweep.py 18749 32768 --quick > $OUT/sweep_d2_config3_quick.log 2>&1 || { echo sweep cfg3 failed; tail -5 $OUT/sweep_d2_config3_quick.log; exit 1; }
tail -14 $OUT/sweep_d2_config3_quick.log
bash experiments/fused64k_policy/pmc.sh $OUT/pmc_cfg3 18749 32768 tiled 3:2:2:0 1:2:17:0 > $OUT/pmc_d2_config3.log 2>&1 || { echo pmc failed; tail $OUT/pmc_d2_config3.log; exit 1; }
cat $OUT/pmc_d2_config3.log
