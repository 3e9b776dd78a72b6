#!/bin/bash
# chirp-z (frame lengths that are not a power of two) in one kernel for M <= 16384: GPU suite, then the sweep at such lengths
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r05_exp16
mkdir -p $OUT
cd $ROOT
timeout -k 10 700 python3 -m pytest tests -m gpu -q > $OUT/pytest.out 2>&1; tail -3 $OUT/pytest.out | tee $OUT/log.txt
timeout -k 10 300 python3 tools/size_sweep.py 7 100 1000 1023 1025 3000 4095 5000 8192 8191 10000 100000 2>&1 | tee -a $OUT/log.txt
