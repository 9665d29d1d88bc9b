#!/bin/bash
set -u
OUT=/root/repo/gpurun_out; mkdir -p $OUT; cd /root/repo
python -m pytest tests -m gpu -q --timeout 900 > $OUT/pytest.log 2>&1; tail -15 $OUT/pytest.log
: > $OUT/exp2.jsonl
show='import sys, json
for l in sys.stdin:
    try: d = json.loads(l)
    except Exception: print(l[:300]); continue
    r = d["roofline"]; print(r["kernel"], "ms/step %.4f" % d["ms_per_step"], "kernel_ms %.4f" % r["kernel_ms_avg"], "GB/s %.0f" % r["achieved"], "frac %.3f" % r["frac"], "samples/s %.3e" % d["value"])'
for fd in 1 0; do
for tile in 0 128 256; do
 for v in "static=1,f=8,cpl=1" "static=1,f=8,cpl=2" "static=1,f=8,cpl=4" "static=1,f=16,cpl=2" "static=0,f=8"; do
  echo -n "fastdiv=$fd tile=$tile $v : "
  DSPFX_FAST_DIV=$fd DSPFX_VARIANT="$v" python bench.py --steps 100 --warmup 200 --no-cpu-baseline --tile $tile 2>>$OUT/exp2.log | tee -a $OUT/exp2.jsonl | python -c "$show"
 done
done
done
