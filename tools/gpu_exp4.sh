#!/bin/bash
set -u
OUT=/root/repo/gpurun_out; mkdir -p $OUT; cd /root/repo
show='import sys, json
for l in sys.stdin:
    try: d = json.loads(l)
    except Exception: print(l[:300]); continue
    r = d["roofline"]; print(r["kernel"], "ms/step %.4f" % d["ms_per_step"], "kernel_ms %.4f" % r["kernel_ms_avg"], "GB/s %.0f" % r["achieved"], "frac %.3f" % r["frac"], "samples/s %.3e" % d["value"])'
for tile in 0 256; do
  echo -n "cfg5 tile=$tile : "; python bench.py --steps 200 --warmup 200 --no-cpu-baseline --tile $tile 2>>$OUT/exp4.log | python -c "$show"
  for f in 4 8 16; do
  echo -n "copy tile=$tile f=$f link0: "; DSPFX_VARIANT="f=$f" python bench.py --config copy --link-flags 0 --no-mix --steps 200 --warmup 50 --no-cpu-baseline --tile $tile 2>>$OUT/exp4.log | python -c "$show"
  echo -n "delay tile=$tile f=$f link0: "; DSPFX_VARIANT="f=$f" python bench.py --config delay --link-flags 0 --no-mix --steps 200 --warmup 200 --no-cpu-baseline --tile $tile 2>>$OUT/exp4.log | python -c "$show"
  done
done
