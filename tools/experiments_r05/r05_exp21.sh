#!/bin/bash
# flagship: s_setprio 3 around the issue of the sixteen prefetch loads (-DF4K_PRIO=1) against what ships; warm, interleaved
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r05_exp21
mkdir -p $OUT
cd $ROOT
summ() { python3 -c "
import json,sys
l=json.loads(sys.stdin.read())
print('%.4f  frac %.4f  2:1 probe %7.1f  kernel/probe %.4f  parity %.3e  probe_ms %s chosen %d' % (l['launch_ms']['median'], l['roofline']['frac'], l['roofline']['measured_copy_GBps'], l['roofline']['frac_of_measured_copy'], l['parity_max_rel_err'], l['placement']['probe_ms'], l['placement']['chosen']))"; }
for r in 1 2 3 4; do for v in base prio; do
  lib=$ROOT/sdr-iq-visualizer_amd/lib_$v/libsdrk.so; [ $v = base ] && lib=$ROOT/sdr-iq-visualizer_amd/lib/libsdrk.so
  echo -n "$v hann: " | tee -a $OUT/log.txt
  SDRK_LIB=$lib timeout -k 10 200 python3 bench.py --no-secondary --cpu-seconds 0 --parity-frames 256 --placement-candidates 6 2>/dev/null | tail -1 | summ | tee -a $OUT/log.txt
done; done
echo done | tee -a $OUT/log.txt
