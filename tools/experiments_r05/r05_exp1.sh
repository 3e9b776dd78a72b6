#!/bin/bash
# round 5, experiment 1 (GPU box): col-pass variants of config 5 — parity subset, A/B, chunk sweep, steady-state evidence
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r05_exp1
mkdir -p $OUT
cd $ROOT
step() {  # name, timeout, command...
    local name=$1 to=$2; shift 2
    echo "== $name" | tee -a $OUT/log.txt
    timeout -k 10 $to "$@" > $OUT/$name.out 2> $OUT/$name.err
    local rc=$?
    echo "rc=$rc" | tee -a $OUT/log.txt
    if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "timeout in $name: stopping" | tee -a $OUT/log.txt; exit 1; fi
    return $rc
}
K="largest_frames or golden_large or staged_col or every_power or sharding_kernels or randomised_large or persistent_grids or channel_bank or stft_large"
for v in e1 e2; do
    SDRK_LIB=$ROOT/sdr-iq-visualizer_amd/lib_$v/libsdrk.so step pytest_$v 420 python3 -m pytest tests/test_parity_gpu.py -q -x -m gpu -k "$K"
    tail -3 $OUT/pytest_$v.out | tee -a $OUT/log.txt
done
step ab_192 500 python3 tools/ab_cfg.py --rounds 3 --cfg 5 base e1 e2; cat $OUT/ab_192.out | tee -a $OUT/log.txt
step ab_128 400 python3 tools/ab_cfg.py --rounds 2 --cfg 5 --env SDRK_SCRATCH_MB=128 base e1 e2; cat $OUT/ab_128.out | tee -a $OUT/log.txt
step ab_256 400 python3 tools/ab_cfg.py --rounds 2 --cfg 5 --env SDRK_SCRATCH_MB=256 base e1 e2; cat $OUT/ab_256.out | tee -a $OUT/log.txt
step steady_plain 120 python3 tools/cfg_steady.py 1048576 256 1048576 hann --transforms 30 --out $OUT/steady_plain.json
step steady_isolated 120 python3 tools/cfg_steady.py 1048576 256 1048576 hann --transforms 30 --isolated --out $OUT/steady_isolated.json
step steady_cold 120 python3 tools/cfg_steady.py 1048576 256 1048576 hann --transforms 30 --warm-ms 0 --out $OUT/steady_cold.json
cd /tmp && export TMPDIR=/tmp
step steady_traced 300 rocprofv3 --kernel-trace --output-format csv -d $OUT/steady_trace -- python3 $ROOT/tools/cfg_steady.py 1048576 256 1048576 hann --transforms 30 --out $OUT/steady_traced.json
step steady_traced_iso 300 rocprofv3 --kernel-trace --output-format csv -d $OUT/steady_trace_iso -- python3 $ROOT/tools/cfg_steady.py 1048576 256 1048576 hann --transforms 30 --isolated --out $OUT/steady_traced_iso.json
cd $ROOT
python3 tools/summarise_cfg_trace.py $OUT/steady_trace $OUT/steady_traced.json > $OUT/steady_trace_summary.json 2>> $OUT/log.txt
python3 tools/summarise_cfg_trace.py $OUT/steady_trace_iso $OUT/steady_traced_iso.json > $OUT/steady_trace_iso_summary.json 2>> $OUT/log.txt
for f in steady_plain steady_isolated steady_cold steady_traced steady_traced_iso; do echo "-- $f"; cat $OUT/$f.out | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['median_ms'], d['min_ms'], d['max_ms'], d['each_ms'][:6], d['each_ms'][-3:], d['telemetry'])"; done | tee -a $OUT/log.txt
echo done | tee -a $OUT/log.txt
