OUT=$PWD/gpurun_out/r06_d; ROOT=$PWD; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/channel_trace -- python3 $ROOT/tools/channel_probe.py --batch 24 > $OUT/channel_probe_traced.json 2> /dev/null
cd $ROOT; python3 tools/summarise_channel_trace.py $OUT/channel_trace --batches 11 --runs 6 > $OUT/channel_trace_summary.json
python3 - <<PY
import json
r=json.load(open("gpurun_out/r06_d/channel_trace_summary.json"))
for x in r["runs"][-3:]: print(x)
import csv,glob
f=glob.glob("gpurun_out/r06_d/channel_trace/*/*kernel_trace.csv")[0]
rows=sorted(csv.DictReader(open(f)), key=lambda r:int(r["Start_Timestamp"]))
rows=[r for r in rows if "pass" in r["Kernel_Name"] or "decimate" in r["Kernel_Name"]][-37:]
t0=int(rows[0]["Start_Timestamp"])
for r in rows: print(r["Kernel_Name"][6:30], r.get("Queue_Id"), round((int(r["Start_Timestamp"])-t0)/1e3,1), round((int(r["End_Timestamp"])-t0)/1e3,1))
PY
rm -rf $OUT/channel_trace
