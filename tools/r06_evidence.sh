#!/bin/bash
# Round-6 evidence on ONE box, final build: the driver's own command three times (once under rocprofv3 --kernel-trace --stats),
# the paced real-time run under the kernel trace, the counter passes of the chain kernels (cfg5 / cfg3 / cfg2) and of config 4 on
# ALL its sweeps (packed two-part f16 = default, round 4's unpacked form, bf16 x 3, f32) -- so that every entry of
# profiles/traffic.json comes from this build.  Everything lands under gpurun_out/r06ev/; summaries are copied into profiles/.
set -u
ROOT=/root/repo
export DSPFX_ROUND=r06
OUT=$ROOT/gpurun_out/r06ev; rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
python3 $ROOT/bench.py --gpus 1 --steps 20 --warmup 5 > $OUT/bench_run1.json 2>$OUT/bench_run1.err
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o drv -- python3 $ROOT/bench.py --gpus 1 --steps 20 --warmup 5 > $OUT/bench_under_rocprof.json 2>$OUT/rocprof.err
python3 $ROOT/bench.py --gpus 1 --steps 20 --warmup 5 > $OUT/bench_run2.json 2>$OUT/bench_run2.err
T=$(find $OUT/trace -name "*kernel_trace.csv" | head -1)
python3 $ROOT/tools/trace_phases.py $T $OUT/bench_under_rocprof.json $OUT/timed_regions.json > $OUT/timed_regions.txt 2>&1
cp $(find $OUT/trace -name "*kernel_stats.csv" | head -1) $OUT/kernel_stats.csv
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/paced_trace -o paced -- python3 $ROOT/bench.py --gpus 1 --steps 20 --warmup 5 --paced > $OUT/bench_paced_under_rocprof.json 2>$OUT/paced_rocprof.err
python3 - <<PY > $OUT/paced_kernel.txt 2>&1
import csv, glob, statistics
f = glob.glob("$OUT/paced_trace/**/*kernel_trace.csv", recursive=True)[0]
rows = [r for r in csv.DictReader(open(f)) if "chain_kernel" in r["Kernel_Name"]]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
d, last_end = [], None
for r in rows:                      # the paced launches are the ones whose predecessor ended more than 1 ms earlier
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    if last_end is not None and s - last_end > 1_000_000:
        d.append((e - s) / 1e3)
    last_end = e
d.sort()
print("paced launches (gap before > 1 ms): %d  kernel us: min %.1f  p50 %.1f  p99 %.1f  max %.1f  mean %.1f" % (
    len(d), d[0], d[len(d) // 2], d[int(len(d) * 0.99)], d[-1], statistics.mean(d)))
PY
rm -rf $OUT/trace/*/*.db $OUT/paced_trace/*/*.db 2>/dev/null
rm -rf $OUT/paced_trace
find $OUT/trace -name "*kernel_trace.csv" -size +20M -delete
cd $ROOT
bash tools/pmc_chain.sh > $OUT/pmc.txt 2>&1
tail -3 $OUT/timed_regions.txt; cat $OUT/paced_kernel.txt; tail -4 $OUT/pmc.txt
fir() { # tag, kernel, output name, extra env...
  tag=$1; kern=$2; name=$3; shift 3
  env "$@" bash tools/fir_pmc.sh $tag > $OUT/firpmc_$tag.txt 2>&1
  python3 tools/fir_pmc_report.py gpurun_out/firpmc_$tag r06 --kernel $kern > $OUT/$name 2>$OUT/firpmc_${tag}_report.err
  find gpurun_out/firpmc_$tag -name "*kernel_trace.csv" -size +8M -delete 2>/dev/null
  find gpurun_out/firpmc_$tag -name "*.db" -delete 2>/dev/null
  head -8 $OUT/$name
}
fir r06halfp fir_halfp_kernel r06_fir_halfp_pmc.json DSPFX_NOP=1
fir r06half fir_half_kernel r06_fir_half_unpacked_pmc.json DSPFX_FIR_PACKED=0
fir r06split fir_split_kernel r06_fir_split_pmc.json DSPFX_FIR_HALF=0
fir r06f32 fir_skew_kernel r06_fir_pmc.json DSPFX_FIR_SPLIT=0
find gpurun_out/pmc_chain -name "*.db" -delete 2>/dev/null
# gpurun brings back at most 64 MiB: keep the summaries, drop the raw counter / trace files they were made from
mkdir -p $OUT/pmc_chain_profiles && cp gpurun_out/pmc_chain/profiles/* $OUT/pmc_chain_profiles/ 2>/dev/null
for d in gpurun_out/firpmc_r06halfp gpurun_out/firpmc_r06half gpurun_out/firpmc_r06split gpurun_out/firpmc_r06f32; do
  mkdir -p $OUT/$(basename $d); cp $d/*_bench.json $d/trace_bench.json $OUT/$(basename $d)/ 2>/dev/null; rm -rf $d
done
rm -rf gpurun_out/pmc_chain
find $OUT/trace -name "*kernel_trace.csv" -delete 2>/dev/null
find $OUT -name "*.db" -delete 2>/dev/null
du -sh gpurun_out | tail -1
python3 - <<PY
import json
for f in ("bench_run1", "bench_under_rocprof", "bench_run2"):
    try:
        d = json.loads(open("$OUT/%s.json" % f).read().strip().splitlines()[-1])
        oc = d.get("other_configs", {})
        print(f, "ms/step %.4f frac %.3f by_step %.3f fixed %.0f us |" % (d["ms_per_step"], d["roofline"]["frac"], d["roofline"].get("frac_by_step", 0), d.get("region_fixed_cost_us") or 0),
              " ".join("%s %.4f/%.4f" % (k, v.get("ms_per_step", 0), (v.get("roofline") or {}).get("kernel_ms_avg", 0)) for k, v in oc.items()),
              "| paced", d.get("paced", {}).get("latency_ms"), "line bytes", len(json.dumps(d)))
    except Exception as ex:
        print(f, "unreadable:", ex)
PY
# the driver's N = 2 command shape WITHOUT a launcher in front of it (bench.py starts its own ranks), both ranks on this box's one GPU
cd $ROOT
DSPFX_BENCH_SHARE_GPU=1 DSPFX_BENCH_COMM=abi timeout 900 python3 bench.py --gpus 2 --steps 20 --warmup 5 > $OUT/two_ranks_self_launched.json 2> $OUT/two_ranks_self_launched.err
echo "two ranks rc=$?"; python3 -c "
import json; d = json.loads(open('$OUT/two_ranks_self_launched.json').read().strip().splitlines()[-1]); print(d['scaling_forms'], d['bus_exchange'])"
