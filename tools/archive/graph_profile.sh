#!/bin/bash
# rocprofv3 kernel-trace stats for the whole-graph kernel (tools/graph_speed.py diamond at 1 048 576 channels):
# the generated kernel's average duration next to the run-by-run kernels'.
set -u
OUT=/root/repo/gpurun_out/prof_graph; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o diamond -- python3 /root/repo/tools/graph_speed.py diamond > $OUT/speed.log 2>$OUT/trace.err
cat $OUT/speed.log | grep -v "^run"
f=$(find $OUT/trace -name "*kernel_stats.csv" | head -1)
python3 - "$f" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: -float(r["TotalDurationNs"]))
print("kernel,calls,avg_us,total_ms,pct")
for r in rows[:12]:
    print("%s,%s,%.1f,%.2f,%s" % (r["Name"][:90], r["Calls"], float(r["AverageNs"]) / 1e3, float(r["TotalDurationNs"]) / 1e6, r["Percentage"]))
PY
