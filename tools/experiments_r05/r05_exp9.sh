#!/bin/bash
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r05_exp9
mkdir -p $OUT
cd $ROOT
timeout -k 10 500 python3 -m pytest tests -m gpu -q > $OUT/pytest.out 2>&1; tail -3 $OUT/pytest.out | tee -a $OUT/log.txt
for seed in 901 902 903; do
timeout -k 10 300 python3 tools/stress_features.py 900 $seed > $OUT/sf_$seed.out 2>&1; tail -1 $OUT/sf_$seed.out | cut -c1-160 | tee -a $OUT/log.txt
timeout -k 10 300 python3 tools/stress_fused.py 500 $seed > $OUT/sfu_$seed.out 2>&1; tail -1 $OUT/sfu_$seed.out | cut -c1-160 | tee -a $OUT/log.txt
done
echo done | tee -a $OUT/log.txt
