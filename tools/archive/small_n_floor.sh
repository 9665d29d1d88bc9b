#!/bin/bash
# Few channels: time-sliced kernel at one / two channels per lane, the standard kernel, and the copy / delay-only floors.
cd /tmp
run() { python3 /root/repo/bench.py --no-cpu-baseline --no-others "$@" --steps 200 --warmup 50 2>/dev/null | python3 -c "
import sys,json
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('%-60s us/step %.2f kern %.2f frac %.3f'%(sys.argv[1],d['ms_per_step']*1e3,d['roofline']['kernel_ms_avg']*1e3,d['roofline']['frac']))" "$VAR $*"; }
for cfg in cfg2 cfg5; do
for n in 16384 32768 49152 65536 81920 131072; do
for v in "ts=1,cpl=1" "ts=1,cpl=2" "ts=0"; do export VAR="[$v]"; export DSPFX_VARIANT="$v"; run --config $cfg --channels $n; done
done
done
