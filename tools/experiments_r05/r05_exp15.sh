#!/bin/bash
# flagship: pass-1 twiddles W4096^(tid k) as powers of a per-thread base (pow_tree, as the LDS-family kernels do) instead of two LDS
# table reads + a product per k: -16 LDS instructions and 162 -> 126 VGPRs.  Parity, then headline A/B, warm, interleaved.
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r05_exp15
mkdir -p $OUT
cd $ROOT
PT=$ROOT/sdr-iq-visualizer_amd/lib_powtree/libsdrk.so
BASE=$ROOT/sdr-iq-visualizer_amd/lib/libsdrk.so
SDRK_LIB=$PT timeout -k 10 700 python3 -m pytest tests -m gpu -q > $OUT/pytest_powtree.out 2>&1; tail -3 $OUT/pytest_powtree.out | tee $OUT/log.txt
summ() { python3 -c "
import json,sys
l=json.loads(sys.stdin.read())
print('%.4f  frac %.4f  2:1 probe %7.1f  kernel/probe %.4f  parity %.3e  rect %.4f  probe_ms %s chosen %d' % (l['launch_ms']['median'], l['roofline']['frac'], l['roofline']['measured_copy_GBps'], l['roofline']['frac_of_measured_copy'], l['parity_max_rel_err'], l['secondary']['config2_other_window']['ms'] if 'secondary' in l and l['secondary'] else -1, l['placement']['probe_ms'], l['placement']['chosen']))"; }
for r in 1 2 3 4; do for v in base powtree; do
  lib=$PT; [ $v = base ] && lib=$BASE
  echo -n "$v hann: " | tee -a $OUT/log.txt
  SDRK_LIB=$lib timeout -k 10 200 python3 bench.py --no-secondary --cpu-seconds 0 --parity-frames 1024 --placement-candidates 6 2>/dev/null | tail -1 | summ | tee -a $OUT/log.txt
done; done
echo done | tee -a $OUT/log.txt
