#!/bin/bash
# gpurun call 8: scalar polling + one flag per workgroup (FU_SPOLL), s_sleep 0 / 1 / 4 between polls, ring depth 1 (and 2);
# phase trace of the sleep-1 build.
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r06_fused8; mkdir -p $OUT
cd $ROOT
for v in spoll0_d1 spoll1_d1 spoll4_d1 spoll1; do
  export SDRK_LIB=$ROOT/sdr-iq-visualizer_amd/lib_fuexp_$v/libsdrk.so
  timeout -k 10 200 python3 experiments/fused64k_policy/sweep.py 4096 65536 --few > $OUT/sweep_$v.log 2>&1 || { echo $v failed; tail -5 $OUT/sweep_$v.log; exit 1; }
  echo "== $v"; tail -7 $OUT/sweep_$v.log
done
export SDRK_LIB=$ROOT/sdr-iq-visualizer_amd/lib_fuexp_spoll1_d1/libsdrk.so
timeout -k 10 200 python3 experiments/fused64k_policy/sweep.py 18749 32768 --few > $OUT/sweep_cfg3_spoll1_d1.log 2>&1 || { echo cfg3 failed; exit 1; }
tail -7 $OUT/sweep_cfg3_spoll1_d1.log
export SDRK_LIB=$ROOT/sdr-iq-visualizer_amd/lib_fuexp_spoll_trace_d1/libsdrk.so
python3 experiments/fused64k_policy/trace.py 4096 65536 > $OUT/trace_spoll_packed.log 2>&1; head -34 $OUT/trace_spoll_packed.log
