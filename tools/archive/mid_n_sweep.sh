#!/bin/bash
# 5-node chain at channel counts between the small-engine range and the headline: default kernel choice vs forced shapes.
cd /tmp
run() { python3 /root/repo/bench.py --no-cpu-baseline --no-others "$@" --steps 100 --warmup 30 2>/dev/null | python3 -c "
import sys,json
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('%-44s kern %8.2f us frac %.3f  %s'%(sys.argv[1],d['roofline']['kernel_ms_avg']*1e3,d['roofline']['frac'], d['roofline']['kernel']))" "$VAR $*"; }
for n in 131072 196608 262144 393216 524288 786432; do
 for v in "" "f=8,cpl=1" "f=8,cpl=2" "f=16,cpl=1"; do export VAR="[$v]"; if [ -n "$v" ]; then export DSPFX_VARIANT="$v"; else unset DSPFX_VARIANT; fi; run --config cfg5 --channels $n; done
done
