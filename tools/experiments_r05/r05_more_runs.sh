summ() { python3 -c "
import json,sys
l=json.loads(sys.stdin.read()); t=l['telemetry']
print('%.4f  %.4f  %.4f  %.4f  %7.1f  %.4f  sclk_after %s  probe_ms %s chosen %d' % (l['launch_ms']['median'], l['launch_ms']['min'], l['launch_ms']['max'], l['roofline']['frac'], l['roofline']['measured_copy_GBps'], l['roofline']['frac_of_measured_copy'], t['after']['sclk_mhz'], l['placement']['probe_ms'], l['placement']['chosen']))"; }
mkdir -p gpurun_out/r05_more
{ echo "# another box, same commands: 6 tuned processes, then 3 with --placement-candidates 1";
  for i in 1 2 3 4 5 6; do python3 bench.py --no-secondary --cpu-seconds 0 --parity-frames 0 2>/dev/null | tail -1 | summ; done;
  for i in 1 2 3; do python3 bench.py --no-secondary --cpu-seconds 0 --parity-frames 0 --placement-candidates 1 2>/dev/null | tail -1 | summ; done; } > gpurun_out/r05_more/runs.txt
cat gpurun_out/r05_more/runs.txt | cut -c1-120
