#!/bin/bash
set -u
OUT=/root/repo/gpurun_out; mkdir -p $OUT; cd /root/repo
python -m pytest tests -m gpu -q --timeout 900 -s -k fast_constant > $OUT/pytest.log 2>&1; grep -E "failing|passed|failed" $OUT/pytest.log
show='import sys, json
for l in sys.stdin:
    try: d = json.loads(l)
    except Exception: print(l[:300]); continue
    r = d["roofline"]; print(r["kernel"], "ms/step %.4f" % d["ms_per_step"], "gpu_ev/step %.4f" % d["gpu_event_ms_per_step"], "kernel_ms %.4f" % r["kernel_ms_avg"], "GB/s %.0f" % r["achieved"], "frac %.3f" % r["frac"], "samples/s %.3e" % d["value"])'
for tile in 0 256; do
 for v in "static=1,f=8,cpl=1" "static=1,f=8,cpl=2"; do
  echo -n "tile=$tile $v : "
  DSPFX_VARIANT="$v" python bench.py --steps 200 --warmup 200 --no-cpu-baseline --tile $tile 2>>$OUT/exp3.log | python -c "$show"
  echo -n "   nomix: "
  DSPFX_VARIANT="$v" python bench.py --steps 200 --warmup 200 --no-cpu-baseline --tile $tile --no-mix 2>>$OUT/exp3.log | python -c "$show"
 done
done
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $OUT/prof_r01 -o cfg5 -- python3 /root/repo/bench.py --steps 100 --warmup 200 --no-cpu-baseline > $OUT/prof_bench.log 2>&1
tail -2 $OUT/prof_bench.log | cut -c1-300
ls -R $OUT/prof_r01 | head -20
