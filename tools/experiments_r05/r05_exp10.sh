#!/bin/bash
# the log epilogue without its square root where no bin of the wave is near the additive floor (-DSDRK_FASTLOG=1): parity, headline A/B, f1 A/B
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r05_exp10
mkdir -p $OUT
cd $ROOT
FL=$ROOT/sdr-iq-visualizer_amd/lib_fastlog/libsdrk.so
SDRK_LIB=$FL timeout -k 10 600 python3 -m pytest tests -m gpu -q -x > $OUT/pytest_fastlog.out 2>&1; tail -4 $OUT/pytest_fastlog.out | tee -a $OUT/log.txt
summ() { python3 -c "
import json,sys
l=json.loads(sys.stdin.read())
print('%.4f  frac %.4f  2:1 probe %7.1f  kernel/probe %.4f  parity %.2e  probe_ms %s chosen %d' % (l['launch_ms']['median'], l['roofline']['frac'], l['roofline']['measured_copy_GBps'], l['roofline']['frac_of_measured_copy'], l['parity_max_rel_err'], l['placement']['probe_ms'], l['placement']['chosen']))"; }
for r in 1 2 3; do for v in base fastlog; do
  lib=$ROOT/sdr-iq-visualizer_amd/lib_$v/libsdrk.so; [ $v = base ] && lib=$ROOT/sdr-iq-visualizer_amd/lib/libsdrk.so
  echo -n "$v hann: " | tee -a $OUT/log.txt
  SDRK_LIB=$lib timeout -k 10 200 python3 bench.py --no-secondary --cpu-seconds 0 --parity-frames 256 --placement-candidates 6 2>/dev/null | tail -1 | summ | tee -a $OUT/log.txt
done; done
for r in 1 2; do for v in base fastlog; do
  lib=$ROOT/sdr-iq-visualizer_amd/lib_$v/libsdrk.so; [ $v = base ] && lib=$ROOT/sdr-iq-visualizer_amd/lib/libsdrk.so
  SDRK_LIB=$lib timeout -k 10 200 python3 tools/feat_probe.py > $OUT/feat_${v}_$r.out 2>&1
  echo "$v: $(grep -E 'warm|transform only' $OUT/feat_${v}_$r.out | sed 's/fused //' | tr '\n' ' ')" | tee -a $OUT/log.txt
done; done
echo done | tee -a $OUT/log.txt
