#!/bin/bash
set -u
OUT=/root/repo/gpurun_out/r02g; mkdir -p $OUT
cd /root/repo
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "time_sliced or golden or chain3 or config or smoke or mixpipe or pipelined" > $OUT/pytest_ts.log 2>&1; echo "pytest rc $?"; tail -6 $OUT/pytest_ts.log
cd /tmp && export TMPDIR=/tmp
B="python3 /root/repo/bench.py --no-cpu-baseline"
for v in "ts=1" "ts=0" "ts=1,cpl=2"; do
  DSPFX_VARIANT="$v" $B --config cfg2 --steps 200 --warmup 50 > "$OUT/cfg2_$v.json" 2>"$OUT/cfg2_$v.err"
  DSPFX_VARIANT="$v" $B --config cfg2 --channels 131072 --steps 200 --warmup 50 > "$OUT/cfg2_128k_$v.json" 2>/dev/null
  DSPFX_VARIANT="$v" $B --config cfg2 --channels 16384 --steps 200 --warmup 50 > "$OUT/cfg2_16k_$v.json" 2>/dev/null
done
DSPFX_VARIANT="ts=1" rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace_ts -o cfg2 -- $B --config cfg2 --steps 100 --warmup 20 > $OUT/trace_ts.json 2>$OUT/trace_ts.err
for m in "" "DSPFX_BENCH_COMM_EARLY=1" "DSPFX_BENCH_NIN=1" "DSPFX_BENCH_COMM=torch"; do
  env $m DSPFX_BENCH_FORCE_DIST=1 $B --steps 200 --warmup 50 --no-others > "$OUT/fd_${m:-late}.json" 2>"$OUT/fd_${m:-late}.err"
done
for f in $OUT/cfg2_*.json $OUT/fd_*.json; do python3 - "$f" <<'PY'
import json,sys
lines=[l for l in open(sys.argv[1]).read().splitlines() if l.startswith('{')]
if not lines: print(sys.argv[1],'NO LINE'); sys.exit()
d=json.loads(lines[-1])
print(sys.argv[1].split('/')[-1], 'ms/step %.4f kern %.4f frac %.3f settle %s %s'%(d['ms_per_step'], d['roofline']['kernel_ms_avg'], d['roofline']['frac'], d['config']['settle']['ms_per_step'], d['config']['plan'][0][:90]))
PY
done
grep "chain_ts\|chain_kernel" $OUT/trace_ts/cfg2_kernel_stats.csv | cut -c1-200
