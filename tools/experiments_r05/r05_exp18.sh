#!/bin/bash
# the scratch chunk size, fine steps around the default (192 MiB), warm: config 5 (8 MiB per frame) and config 3 (0.5 MiB per frame)
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r05_exp18
mkdir -p $OUT
cd $ROOT
echo "== config 5" | tee $OUT/log.txt
timeout -k 10 500 python3 tools/ab_steady.py --rounds 3 --cfg "1048576 256 1048576 hann" base base:SDRK_SCRATCH_MB=160 base:SDRK_SCRATCH_MB=176 base:SDRK_SCRATCH_MB=208 base:SDRK_SCRATCH_MB=224 base:SDRK_SCRATCH_MB=240 2>&1 | tail -8 | tee -a $OUT/log.txt
echo "== config 3" | tee -a $OUT/log.txt
timeout -k 10 500 python3 tools/ab_steady.py --rounds 3 --cfg "65536 18749 32768 hann" base base:SDRK_SCRATCH_MB=160 base:SDRK_SCRATCH_MB=176 base:SDRK_SCRATCH_MB=208 base:SDRK_SCRATCH_MB=224 base:SDRK_SCRATCH_MB=240 2>&1 | tail -8 | tee -a $OUT/log.txt
echo done | tee -a $OUT/log.txt
