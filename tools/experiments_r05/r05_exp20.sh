#!/bin/bash
# flagship: two frames prefetched (F4K_DEPTH=2) now that power-tree twiddles leave the registers for it (126 -> 140 VGPRs), against
# power-tree alone and what ships.  The copy probe likes depth 2 at three workgroups per CU (+2 %); round 2 measured the transform 2.5 % slower with it.
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r05_exp20
mkdir -p $OUT
cd $ROOT
summ() { python3 -c "
import json,sys
l=json.loads(sys.stdin.read())
print('%.4f  frac %.4f  2:1 probe %7.1f  kernel/probe %.4f  parity %.3e  probe_ms %s chosen %d' % (l['launch_ms']['median'], l['roofline']['frac'], l['roofline']['measured_copy_GBps'], l['roofline']['frac_of_measured_copy'], l['parity_max_rel_err'], l['placement']['probe_ms'], l['placement']['chosen']))"; }
for r in 1 2 3 4; do for v in base pt ptd2; do
  lib=$ROOT/sdr-iq-visualizer_amd/lib_$v/libsdrk.so; [ $v = base ] && lib=$ROOT/sdr-iq-visualizer_amd/lib/libsdrk.so
  echo -n "$v hann: " | tee -a $OUT/log.txt
  SDRK_LIB=$lib timeout -k 10 200 python3 bench.py --no-secondary --cpu-seconds 0 --parity-frames 1024 --placement-candidates 6 2>/dev/null | tail -1 | summ | tee -a $OUT/log.txt
done; done
SDRK_LIB=$ROOT/sdr-iq-visualizer_amd/lib_ptd2/libsdrk.so timeout -k 10 700 python3 -m pytest tests -m gpu -q > $OUT/pytest_ptd2.out 2>&1; tail -3 $OUT/pytest_ptd2.out | tee -a $OUT/log.txt
echo done | tee -a $OUT/log.txt
