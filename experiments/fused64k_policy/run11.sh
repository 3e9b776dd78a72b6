#!/bin/bash
# gpurun call 11: progressive hand-over (FU_PROG), ring depth 1 and 2, packed and config-3 shape; phase trace.
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r06_fused11; mkdir -p $OUT
cd $ROOT
for v in flagl2_d1 prog_d1 prog; do
  export SDRK_LIB=$ROOT/sdr-iq-visualizer_amd/lib_fuexp_$v/libsdrk.so
  timeout -k 10 200 python3 experiments/fused64k_policy/sweep.py 4096 65536 --few > $OUT/sweep_$v.log 2>&1 || { echo $v failed; tail -5 $OUT/sweep_$v.log; exit 1; }
  echo "== $v"; tail -7 $OUT/sweep_$v.log
done
for v in prog_d1 prog; do
export SDRK_LIB=$ROOT/sdr-iq-visualizer_amd/lib_fuexp_$v/libsdrk.so
timeout -k 10 200 python3 experiments/fused64k_policy/sweep.py 18749 32768 --few > $OUT/sweep_cfg3_$v.log 2>&1 || { echo cfg3 failed; exit 1; }
echo "== cfg3 $v"; tail -7 $OUT/sweep_cfg3_$v.log
done
export SDRK_LIB=$ROOT/sdr-iq-visualizer_amd/lib_fuexp_prog_trace_d1/libsdrk.so
python3 experiments/fused64k_policy/trace.py 4096 65536 > $OUT/trace_prog_packed.log 2>&1; head -20 $OUT/trace_prog_packed.log
