#!/bin/bash
# layout experiment: frame-major vs channel-tiled, per variant
set -u
OUT=/root/repo/gpurun_out; mkdir -p $OUT; cd /root/repo
python -m pytest tests -m gpu -q --timeout 900 > $OUT/pytest.log 2>&1; tail -3 $OUT/pytest.log
: > $OUT/exp1.jsonl
show='import sys, json
for l in sys.stdin:
    try: d = json.loads(l)
    except Exception: print(l[:300]); continue
    r = d["roofline"]; print(r["kernel"], "ms/step %.4f" % d["ms_per_step"], "kernel_ms %.4f" % r["kernel_ms_avg"], "GB/s %.0f" % r["achieved"], "frac %.3f" % r["frac"], "samples/s %.3e" % d["value"])'
for tile in 0 64 128 256 512 1024; do
 for v in "static=1,f=8,cpl=1" "static=1,f=8,cpl=2" "static=1,f=8,cpl=4"; do
  echo -n "tile=$tile $v : "
  DSPFX_VARIANT="$v" python bench.py --steps 100 --warmup 200 --no-cpu-baseline --tile $tile 2>>$OUT/exp1.log | tee -a $OUT/exp1.jsonl | python -c "$show"
 done
done
echo "link0:"; for tile in 0 256; do DSPFX_VARIANT="static=1,f=8,cpl=2" python bench.py --steps 100 --warmup 200 --no-cpu-baseline --tile $tile --link-flags 0 2>>$OUT/exp1.log | python -c "$show"; done
