#!/bin/bash
# N = 4096 kernel: 3 against 4 workgroups per CU (rectangular window: 37 KB of LDS, 128 VGPRs), same bench command, interleaved
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r05_exp7
mkdir -p $OUT
cd $ROOT
summ() { python3 -c "
import json,sys
l=json.loads(sys.stdin.read()); 
print('%.4f  frac %.4f  copy %7.1f  kernel/copy %.4f  copy11 %7.1f  probe_ms %s chosen %d' % (l['launch_ms']['median'], l['roofline']['frac'], l['roofline']['measured_copy_GBps'], l['roofline']['frac_of_measured_copy'], l['roofline']['copy_1to1_GBps'], l['placement']['probe_ms'], l['placement']['chosen']))"; }
for r in 1 2 3; do for v in base w4; do
  lib=$ROOT/sdr-iq-visualizer_amd/lib_$v/libsdrk.so; [ $v = base ] && lib=$ROOT/sdr-iq-visualizer_amd/lib/libsdrk.so
  echo -n "$v rect: " | tee -a $OUT/log.txt
  SDRK_LIB=$lib timeout -k 10 200 python3 bench.py --window rect --no-secondary --cpu-seconds 0 --parity-frames 64 --placement-candidates 6 2>/dev/null | tail -1 | summ | tee -a $OUT/log.txt
done; done
echo done | tee -a $OUT/log.txt
