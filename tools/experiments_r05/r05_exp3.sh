#!/bin/bash
# round 5, GPU call 3: the whole -m gpu suite at HEAD, the continuous-channel leg (plain, traced), the bench line
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r05_exp3
mkdir -p $OUT
cd $ROOT
step() {
    local name=$1 to=$2; shift 2
    echo "== $name" | tee -a $OUT/log.txt
    timeout -k 10 $to "$@" > $OUT/$name.out 2> $OUT/$name.err
    local rc=$?
    echo "rc=$rc" | tee -a $OUT/log.txt
    if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "timeout in $name: stopping" | tee -a $OUT/log.txt; exit 1; fi
    return $rc
}
step pytest_gpu 1000 python3 -m pytest tests -m gpu -x -q --durations=15; tail -25 $OUT/pytest_gpu.out | tee -a $OUT/log.txt
step chan16 120 python3 tools/channel_probe.py --batch 16; cat $OUT/chan16.out | tee -a $OUT/log.txt
step chan24 120 python3 tools/channel_probe.py --batch 24; cat $OUT/chan24.out | tee -a $OUT/log.txt
step chan48 120 python3 tools/channel_probe.py --batch 48; cat $OUT/chan48.out | tee -a $OUT/log.txt
step bench 600 python3 bench.py; cp $OUT/bench.out $OUT/bench_n1.json; python3 -c "
import json,sys; d=json.loads(open('$OUT/bench.out').read().strip().splitlines()[-1]); s=d.get('secondary',{})
print('value',d['value'],'frac',d['roofline']['frac'],'placement',d['placement'])
for k in ('config3','config5_one_channel','config5_one_channel_rect'):
    r=s.get(k,{}); print(k, r.get('ms'), r.get('frac'), r.get('scratch_placement'))
c=s.get('config5_channel_with_ring_and_gather',{}); print('channel', c.get('ms'), c.get('pipelined',{}).get('ms'), c.get('ring_rows_with_maxhold16_companion'), c.get('ring_rows_equal_plain_transform'), c.get('decimated_rows_equal_numpy_max_of_ring_rows'))
print('error', s.get('error')); print('copy', d['roofline'].get('copy_1to1_GBps'), d['roofline'].get('copy_1to1_small_footprint'))
" | tee -a $OUT/log.txt
step feat 300 python3 tools/feat_probe.py; tail -5 $OUT/feat.out | tee -a $OUT/log.txt
cd /tmp && export TMPDIR=/tmp
step chan24_trace 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/chan24_trace -- python3 $ROOT/tools/channel_probe.py --batch 24
echo done | tee -a $OUT/log.txt
