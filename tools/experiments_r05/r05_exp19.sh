#!/bin/bash
# EXPERIMENT: the N = 4096 transform with one wave per frame (fft4096_wave.hip, -DSDRK_F4K_WAVE=1 for batches >= 1024 frames)
# against the workgroup-per-frame flagship: parity through the bench's own check, the GPU suite (the bit comparisons with the feature
# kernel's rows are expected to differ: other twiddle arithmetic), then headline A/B, warm, interleaved
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r05_exp19
mkdir -p $OUT
cd $ROOT
WV=$ROOT/sdr-iq-visualizer_amd/lib_wave/libsdrk.so
BASE=$ROOT/sdr-iq-visualizer_amd/lib/libsdrk.so
summ() { python3 -c "
import json,sys
l=json.loads(sys.stdin.read())
print('%.4f  frac %.4f  2:1 probe %7.1f  kernel/probe %.4f  parity %.3e  rect %s  probe_ms %s chosen %d' % (l['launch_ms']['median'], l['roofline']['frac'], l['roofline']['measured_copy_GBps'], l['roofline']['frac_of_measured_copy'], l['parity_max_rel_err'], l.get('secondary',{}).get('config2_other_window',{}).get('ms'), l['placement']['probe_ms'], l['placement']['chosen']))"; }
for r in 1 2 3; do for v in base wave; do
  lib=$WV; [ $v = base ] && lib=$BASE
  echo -n "$v hann: " | tee -a $OUT/log.txt
  SDRK_LIB=$lib timeout -k 10 200 python3 bench.py --no-secondary --cpu-seconds 0 --parity-frames 1024 --placement-candidates 6 2> $OUT/err_${v}_$r.txt | tail -1 | summ | tee -a $OUT/log.txt
done; done
SDRK_LIB=$WV timeout -k 10 700 python3 -m pytest tests -m gpu -q > $OUT/pytest_wave.out 2>&1; tail -8 $OUT/pytest_wave.out | tee -a $OUT/log.txt
echo done | tee -a $OUT/log.txt
