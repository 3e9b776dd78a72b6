#!/bin/bash
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r05_final
mkdir -p $OUT
cd $ROOT
timeout -k 10 500 python3 -m pytest tests -m gpu -q > $OUT/pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -3 $OUT/pytest_gpu.log
timeout -k 10 600 bash tools/soak.sh 600 1; cp gpurun_out/soak.log $OUT/soak.log; cat $OUT/soak.log
