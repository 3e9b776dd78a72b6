#!/bin/bash
# developer helper: rebuild libsdrk with a ring/lag variant of the fused 65536 kernel, time it and read its PMC traffic
S=$1; L=$2
cd /root/repo/sdr-iq-visualizer_amd/csrc && touch fft_fused64k.hip && make CXXFLAGS="-O3 -std=c++17 -fPIC -Wall -Wno-unused-function -DFU_RING_SLOTS_N=$S -DFU_LAG_N=$L" 2>&1 | grep -E "error" 
cd /root/repo && /usr/local/graft/bin/gpurun --timeout 600 -- 'R=$GRAFT_REPO_ROOT; python3 tools/one_config.py 65536 4096 65536 rect | cut -c1-200; python3 tools/one_config.py 65536 18749 32768 hann | cut -c1-200; cd /tmp; export TMPDIR=/tmp; for c in FETCH_SIZE WRITE_SIZE; do rm -rf $R/gpurun_out/pv_$c; timeout -k 10 120 rocprofv3 --pmc $c --output-format csv -d $R/gpurun_out/pv_$c -- python3 $R/tools/one_config.py 65536 4096 65536 rect > /dev/null 2>&1; done; cd $R; python3 - <<PY
import csv,glob
for c in ("FETCH_SIZE","WRITE_SIZE"):
    for f in glob.glob(f"gpurun_out/pv_{c}/*/*counter_collection.csv"):
        rows=[r for r in csv.DictReader(open(f)) if "fused64k" in r["Kernel_Name"]]
        v=[float(r["Counter_Value"]) for r in rows]
        if v: print(c, sum(v)/len(v)*1024/(4096*65536), "B/sample raw")
PY' 2>&1 | grep -E "one_config|B/sample"
