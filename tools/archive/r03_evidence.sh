#!/bin/bash
# Round-3 evidence on ONE box, final build: the driver's own command three times (once under rocprofv3 --kernel-trace --stats),
# the counter passes, the paced real-time run under the kernel trace, the bus and config-2 A/B tables.
# Everything lands under gpurun_out/r03ev/; summaries are copied into profiles/ by hand (names r03_*).
set -u
ROOT=/root/repo
OUT=$ROOT/gpurun_out/r03ev; rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
python3 $ROOT/bench.py --gpus 1 --steps 20 --warmup 5 > $OUT/bench_run1.json 2>$OUT/bench_run1.err
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o drv -- python3 $ROOT/bench.py --gpus 1 --steps 20 --warmup 5 > $OUT/bench_under_rocprof.json 2>$OUT/rocprof.err
python3 $ROOT/bench.py --gpus 1 --steps 20 --warmup 5 > $OUT/bench_run2.json 2>$OUT/bench_run2.err
T=$(find $OUT/trace -name "*kernel_trace.csv" | head -1)
python3 $ROOT/tools/trace_phases.py $T $OUT/bench_under_rocprof.json $OUT/timed_regions.json > $OUT/timed_regions.txt 2>&1
cp $(find $OUT/trace -name "*kernel_stats.csv" | head -1) $OUT/kernel_stats.csv
# paced run under the kernel trace: one block per 2.667 ms for 5 s, the bus of the same block inside the launch
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/paced_trace -o paced -- python3 $ROOT/bench.py --gpus 1 --steps 20 --warmup 5 --paced > $OUT/bench_paced_under_rocprof.json 2>$OUT/paced_rocprof.err
python3 - <<PY > $OUT/paced_kernel.txt 2>&1
import csv, glob, statistics
f = glob.glob("$OUT/paced_trace/**/*kernel_trace.csv", recursive=True)[0]
rows = [r for r in csv.DictReader(open(f)) if "chain_kernel" in r["Kernel_Name"]]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
# the paced launches are the ones whose predecessor ended more than 1 ms earlier
d, last_end = [], None
for r in rows:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    if last_end is not None and s - last_end > 1_000_000:
        d.append((e - s) / 1e3)
    last_end = e
d.sort()
print("paced launches (gap before > 1 ms): %d  kernel us: min %.1f  p50 %.1f  p99 %.1f  max %.1f  mean %.1f" % (
    len(d), d[0], d[len(d) // 2], d[int(len(d) * 0.99)], d[-1], statistics.mean(d)))
PY
rm -rf $OUT/trace/*/*.db $OUT/paced_trace/*/*.db 2>/dev/null
# keep the merge small: the raw traces stay on the box except the timed-region rows
rm -rf $OUT/paced_trace
find $OUT/trace -name "*kernel_trace.csv" -size +20M -delete
cd $ROOT
bash tools/r03_pmc.sh > $OUT/pmc.txt 2>&1
bash tools/r03_bus_ab.sh > /dev/null 2>&1
bash tools/r03_cfg2_ab.sh > /dev/null 2>&1
tail -3 $OUT/timed_regions.txt; cat $OUT/paced_kernel.txt; tail -4 $OUT/pmc.txt; cat gpurun_out/r03_bus_ab.txt gpurun_out/r03_cfg2_ab.txt
# FIR counter passes of this build (tools/fir_pmc.sh writes under gpurun_out/firpmc_<round>/)
DSPFX_FIR_SPLIT=0 bash tools/fir_pmc.sh r03 > $OUT/firpmc.txt 2>&1
DSPFX_FIR_SPLIT=1 bash tools/fir_pmc.sh r03split > $OUT/firpmc_split.txt 2>&1
find gpurun_out/firpmc_r03 gpurun_out/firpmc_r03split -name "*kernel_trace.csv" -size +8M -delete 2>/dev/null
