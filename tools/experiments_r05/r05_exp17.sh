#!/bin/bash
# the scratch chunk size of the two-pass plans once more, WARM this time (round 4's sweep ran on tools that timed in the clock ramp):
# config 3 and config 5 with SDRK_SCRATCH_MB = 48 ... 384 (default 192), tools/ab_steady.py, two rounds, interleaved
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r05_exp17
mkdir -p $OUT
cd $ROOT
echo "== config 5" | tee $OUT/log.txt
timeout -k 10 500 python3 tools/ab_steady.py --rounds 2 --cfg "1048576 256 1048576 hann" base base:SDRK_SCRATCH_MB=48 base:SDRK_SCRATCH_MB=64 base:SDRK_SCRATCH_MB=96 base:SDRK_SCRATCH_MB=128 base:SDRK_SCRATCH_MB=256 base:SDRK_SCRATCH_MB=384 2>&1 | tail -9 | tee -a $OUT/log.txt
echo "== config 3" | tee -a $OUT/log.txt
timeout -k 10 500 python3 tools/ab_steady.py --rounds 2 --cfg "65536 18749 32768 hann" base base:SDRK_SCRATCH_MB=48 base:SDRK_SCRATCH_MB=64 base:SDRK_SCRATCH_MB=96 base:SDRK_SCRATCH_MB=128 base:SDRK_SCRATCH_MB=256 base:SDRK_SCRATCH_MB=384 2>&1 | tail -9 | tee -a $OUT/log.txt
echo done | tee -a $OUT/log.txt
