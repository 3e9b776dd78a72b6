#!/bin/bash
# Every randomised stress tool with fresh seeds, one after the other (GPU box): tools/soak.sh <first_seed> <rounds>
# One line per finished tool in gpurun_out/soak.log; stops at the first failure.
S=${1:-100}; R=${2:-3}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; cd "$ROOT"; mkdir -p gpurun_out
LOG=gpurun_out/soak.log; : > $LOG
for r in $(seq 0 $((R-1))); do
  seed=$((S + r))
  for job in "stress_features.py 900" "stress_fused.py 500" "stress_host.py 120" "stress_waterfall.py 120" "stress_large.py 30" "stress_fused64k.py 150"; do
    set -- $job
    args="$2 $seed"
    t0=$(date +%s)
    if timeout -k 10 420 python3 tools/$1 $args > gpurun_out/soak_last.log 2>&1; then
      echo "seed $seed $1 ok ($(( $(date +%s) - t0 )) s): $(tail -n 1 gpurun_out/soak_last.log | cut -c1-160)" >> $LOG
    else
      echo "seed $seed $1 FAILED" >> $LOG; tail -n 15 gpurun_out/soak_last.log >> $LOG; exit 1
    fi
  done
done
echo "soak finished" >> $LOG
