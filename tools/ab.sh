#!/bin/bash
# Developer helper (GPU box): time BASELINE configs 3 and 5 with each library variant, interleaved, 2 rounds.
#   tools/ab.sh base linear nostore ...     ("base" = lib/libsdrk.so)
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
for round in 1 2; do
  for v in "$@"; do
    lib=$ROOT/sdr-iq-visualizer_amd/lib_$v/libsdrk.so; [ "$v" = base ] && lib=$ROOT/sdr-iq-visualizer_amd/lib/libsdrk.so
    for cfg in "65536 18749 32768 hann" "1048576 256 1048576 hann"; do
      echo -n "$v r$round: "; SDRK_LIB=$lib python3 $ROOT/tools/one_config.py $cfg | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['nfft'], d['ms'], 'ms', d['algorithmic_GBps'], 'GB/s')"
    done
  done
done
