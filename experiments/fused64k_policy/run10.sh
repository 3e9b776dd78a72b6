#!/bin/bash
# gpurun call 10: pipelined polling (2 / 4 / 8 loads in flight), ring depth 1; phase trace of the 4-deep build.
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r06_fused10; mkdir -p $OUT
cd $ROOT
for v in flagl2_d1 pp2_d1 pp4_d1 pp8_d1 pp4s1_d1; do
  export SDRK_LIB=$ROOT/sdr-iq-visualizer_amd/lib_fuexp_$v/libsdrk.so
  timeout -k 10 200 python3 experiments/fused64k_policy/sweep.py 4096 65536 --few > $OUT/sweep_$v.log 2>&1 || { echo $v failed; tail -5 $OUT/sweep_$v.log; exit 1; }
  echo "== $v"; tail -7 $OUT/sweep_$v.log
done
export SDRK_LIB=$ROOT/sdr-iq-visualizer_amd/lib_fuexp_pp4_d1/libsdrk.so
timeout -k 10 200 python3 experiments/fused64k_policy/sweep.py 18749 32768 --few > $OUT/sweep_cfg3_pp4_d1.log 2>&1 || { echo cfg3 failed; exit 1; }
tail -7 $OUT/sweep_cfg3_pp4_d1.log
export SDRK_LIB=$ROOT/sdr-iq-visualizer_amd/lib_fuexp_pp4_trace_d1/libsdrk.so
python3 experiments/fused64k_policy/trace.py 4096 65536 > $OUT/trace_pp4_packed.log 2>&1; head -20 $OUT/trace_pp4_packed.log
