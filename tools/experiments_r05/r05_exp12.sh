#!/bin/bash
# complex values as packed register pairs (cplx.h, SDRK_PACKED_CF=1, now the default) against the scalar build (lib_scalar):
# bit identity on every frame-length family, then headline / f1 / config 3 / config 5 / size sweep A/B, warm, interleaved
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r05_exp12
mkdir -p $OUT
cd $ROOT
SC=$ROOT/sdr-iq-visualizer_amd/lib_scalar/libsdrk.so
BASE=$ROOT/sdr-iq-visualizer_amd/lib/libsdrk.so
SDRK_LIB=$BASE timeout -k 10 200 python3 tools/hash_outputs.py > $OUT/hash_packed.txt 2> $OUT/hash_packed.err
SDRK_LIB=$SC timeout -k 10 200 python3 tools/hash_outputs.py > $OUT/hash_scalar.txt 2> $OUT/hash_scalar.err
echo "hash lines: $(wc -l < $OUT/hash_packed.txt) packed, $(wc -l < $OUT/hash_scalar.txt) scalar; differing: $(diff $OUT/hash_packed.txt $OUT/hash_scalar.txt | grep -c '^<')" | tee -a $OUT/log.txt
diff $OUT/hash_packed.txt $OUT/hash_scalar.txt | head -20 | tee -a $OUT/log.txt
timeout -k 10 700 python3 -m pytest tests -m gpu -q > $OUT/pytest_subset.out 2>&1; tail -3 $OUT/pytest_subset.out | tee -a $OUT/log.txt
summ() { python3 -c "
import json,sys
l=json.loads(sys.stdin.read())
print('%.4f  frac %.4f  2:1 probe %7.1f  kernel/probe %.4f  parity %.2e  probe_ms %s chosen %d' % (l['launch_ms']['median'], l['roofline']['frac'], l['roofline']['measured_copy_GBps'], l['roofline']['frac_of_measured_copy'], l['parity_max_rel_err'], l['placement']['probe_ms'], l['placement']['chosen']))"; }
for r in 1 2 3; do for v in base scalar; do
  lib=$ROOT/sdr-iq-visualizer_amd/lib_$v/libsdrk.so; [ $v = base ] && lib=$BASE
  echo -n "$v hann: " | tee -a $OUT/log.txt
  SDRK_LIB=$lib timeout -k 10 200 python3 bench.py --no-secondary --cpu-seconds 0 --parity-frames 256 --placement-candidates 6 2>/dev/null | tail -1 | summ | tee -a $OUT/log.txt
done; done
for r in 1 2; do for v in base f1pk scalar; do
  lib=$ROOT/sdr-iq-visualizer_amd/lib_$v/libsdrk.so; [ $v = base ] && lib=$BASE
  SDRK_LIB=$lib timeout -k 10 200 python3 tools/feat_probe.py > $OUT/feat_${v}_$r.out 2>&1
  echo "$v: $(grep -E 'warm|transform only' $OUT/feat_${v}_$r.out | sed 's/fused //' | tr '\n' ' ')" | tee -a $OUT/log.txt
done; done
echo "== config 5" | tee -a $OUT/log.txt
timeout -k 10 300 python3 tools/ab_steady.py --rounds 3 --cfg "1048576 256 1048576 hann" base scalar 2>&1 | tail -4 | tee -a $OUT/log.txt
echo "== config 3" | tee -a $OUT/log.txt
timeout -k 10 300 python3 tools/ab_steady.py --rounds 3 --cfg "65536 18749 32768 hann" base scalar 2>&1 | tail -4 | tee -a $OUT/log.txt
for v in base scalar; do
  lib=$ROOT/sdr-iq-visualizer_amd/lib_$v/libsdrk.so; [ $v = base ] && lib=$BASE
  SDRK_LIB=$lib timeout -k 10 200 python3 tools/size_sweep.py > $OUT/size_sweep_$v.log 2>&1
done
python3 - $OUT/size_sweep_base.log $OUT/size_sweep_scalar.log <<PY | tee -a $OUT/log.txt
import json,sys
rd=lambda p:{json.loads(l)["nfft"]:json.loads(l)["ms"] for l in open(p) if l.startswith("{") and "nfft" in l}
a,b=rd(sys.argv[1]),rd(sys.argv[2])
for n in a: print("N=%8d packed %.4f ms scalar %.4f ms  packed/scalar %.3f" % (n,a[n],b.get(n,float("nan")),a[n]/b.get(n,float("nan"))))
PY
echo done | tee -a $OUT/log.txt
