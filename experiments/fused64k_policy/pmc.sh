#!/bin/bash
# Round-6 gated experiment (GPU box): fabric-side bytes per launch of fused64k_kernel at chosen points of the policy sweep, one
# counter per rocprofv3 pass, the program directly after `--`.  Read bytes = 2 x FETCH_SIZE (KB) x 1024 on gfx950 (the counter
# tallies a 128-byte request as 64, MI355X_MICROARCH.md "HBM"); written bytes = WRITE_SIZE (KB) x 1024.
#   experiments/fused64k_policy/pmc.sh OUTDIR FRAMES HOP "wgpc:in:out:nowait" ...        ("tiled" = the two tiled launches)
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$1; NF=$2; HOP=$3; shift 3
case "$OUT" in /*) ;; *) OUT="$ROOT/$OUT";; esac
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
COUNTERS=${FU_PMC_COUNTERS:-"FETCH_SIZE WRITE_SIZE"}
for pt in "$@"; do
  kind=fused
  if [ "$pt" = tiled ]; then kind=tiled; else
    IFS=: read wg pin pout nw <<< "$pt"
    export SDRK_FU_WG_PER_CU=$wg SDRK_FU_IN_AUX=$pin SDRK_FU_OUT_AUX=$pout SDRK_FU_NOWAIT=$nw
  fi
  for c in $COUNTERS; do
    d=/tmp/fupmc_${pt//:/_}_$c; rm -rf $d
    if ! timeout -k 10 180 rocprofv3 --pmc $c --output-format csv -d $d -- python3 "$ROOT/experiments/fused64k_policy/one_point.py" $NF $HOP $kind 4 \
          > "$OUT/${pt//:/_}_$c.out" 2> "$OUT/${pt//:/_}_$c.err"; then
      echo "[pmc] pass failed: $pt $c — stopping"; tail -3 "$OUT/${pt//:/_}_$c.err"; exit 1
    fi
    python3 - "$pt" "$c" "$d" "$NF" "$HOP" <<'PY'
import csv, glob, sys, collections
pt, c, d, nf, hop = sys.argv[1:6]
acc = collections.defaultdict(list)
for f in glob.glob(d + "/*/*counter_collection.csv"):
    per = collections.defaultdict(float)
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        if r["Counter_Name"] == c and ("fused64k_kernel" in k or "_pass_" in k):
            name = "fused64k" if "fused64k" in k else ("col_pass" if "col_pass" in k else "row_pass")
            per[(r["Dispatch_Id"], name)] += float(r["Counter_Value"])
    for (disp, name), x in per.items():
        acc[name].append(x)
for name, xs in sorted(acc.items()):
    xs = xs[len(xs) // 2:]            # the later half: warm dispatches
    kb = sum(xs) / len(xs)
    gb = kb * 1024 * (2 if c == "FETCH_SIZE" else 1) / 1e9
    print(f"{pt:>14} {c:>10} {name:>9} dispatches {len(xs):3d}  GB per dispatch {gb:8.4f}", flush=True)
PY
    rm -rf $d
  done
done
