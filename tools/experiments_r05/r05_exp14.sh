#!/bin/bash
# after tidying cplx.h (shared butterfly code, add_mi / add_pi in both forms): bit identity packed == scalar again, the GPU suite
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r05_exp14
mkdir -p $OUT
cd $ROOT
SDRK_LIB=$ROOT/sdr-iq-visualizer_amd/lib/libsdrk.so timeout -k 10 200 python3 tools/hash_outputs.py > $OUT/hash_packed.txt 2> $OUT/hash_packed.err
SDRK_LIB=$ROOT/sdr-iq-visualizer_amd/lib_scalar/libsdrk.so timeout -k 10 200 python3 tools/hash_outputs.py > $OUT/hash_scalar.txt 2> $OUT/hash_scalar.err
echo "hash lines: $(wc -l < $OUT/hash_packed.txt) packed, $(wc -l < $OUT/hash_scalar.txt) scalar; differing: $(diff $OUT/hash_packed.txt $OUT/hash_scalar.txt | grep -c '^<')" | tee $OUT/log.txt
timeout -k 10 700 python3 -m pytest tests -m gpu -q > $OUT/pytest.out 2>&1; tail -2 $OUT/pytest.out | tee -a $OUT/log.txt
timeout -k 10 200 python3 bench.py --no-secondary --cpu-seconds 0 2>/dev/null | tail -1 | python3 -c "
import json,sys
l=json.loads(sys.stdin.read()); print('bench', l['value'], l['ms_per_step'], l['roofline']['frac'], l['parity_max_rel_err'])" | tee -a $OUT/log.txt
