#!/bin/bash
set -u
OUT=/root/repo/gpurun_out/r02h; mkdir -p $OUT
cd /root/repo
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "time_sliced or golden or chain3 or config or smoke or mixpipe or pipelined" > $OUT/pytest_ts.log 2>&1; echo "pytest rc $?"; tail -4 $OUT/pytest_ts.log
cd /tmp && export TMPDIR=/tmp
B="python3 /root/repo/bench.py --no-cpu-baseline"
for v in "ts=1" "ts=0"; do
  for n in 16384 32768 65536 131072; do
    DSPFX_VARIANT="$v" $B --config cfg2 --channels $n --steps 200 --warmup 50 > "$OUT/cfg2_${n}_$v.json" 2>"$OUT/cfg2_${n}_$v.err"
  done
  DSPFX_VARIANT="$v" $B --config cfg5 --channels 65536 --steps 200 --warmup 50 --no-others > "$OUT/cfg5_65536_$v.json" 2>/dev/null
  DSPFX_VARIANT="$v" $B --config cfg5 --channels 131072 --steps 200 --warmup 50 --no-others > "$OUT/cfg5_131072_$v.json" 2>/dev/null
done
for f in $OUT/cfg*.json; do python3 - "$f" <<'PY'
import json,sys
lines=[l for l in open(sys.argv[1]).read().splitlines() if l.startswith('{')]
if not lines: print(sys.argv[1],'NO LINE'); sys.exit()
d=json.loads(lines[-1])
print(sys.argv[1].split('/')[-1], 'ms/step %.4f kern %.4f frac %.3f settle %s'%(d['ms_per_step'], d['roofline']['kernel_ms_avg'], d['roofline']['frac'], d['config']['settle']['ms_per_step']))
PY
done
