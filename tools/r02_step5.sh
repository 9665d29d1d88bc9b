#!/bin/bash
set -u
OUT=/root/repo/gpurun_out/r02e; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
B="python3 /root/repo/bench.py --no-cpu-baseline"
$B --config cfg4 --steps 50 --warmup 20 > $OUT/cfg4.json 2>$OUT/cfg4.err
DSPFX_FIR_FUSE=0 $B --config cfg4 --steps 50 --warmup 20 > $OUT/cfg4_nofuse.json 2>$OUT/cfg4_nofuse.err
for m in abi torch; do
  DSPFX_BENCH_COMM=$m DSPFX_BENCH_FORCE_DIST=1 $B --steps 200 --warmup 50 --no-others > $OUT/fd_$m.json 2>$OUT/fd_$m.err
done
DSPFX_BENCH_MIX_BATCH=32 DSPFX_BENCH_FORCE_DIST=1 $B --steps 200 --warmup 50 --no-others > $OUT/fd_abi_b32.json 2>$OUT/fd_abi_b32.err
$B --steps 200 --warmup 50 --no-others > $OUT/plain.json 2>$OUT/plain.err
timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -q -k "fir" > $OUT/pytest_fir.log 2>&1; tail -3 $OUT/pytest_fir.log
for f in cfg4 cfg4_nofuse fd_abi fd_torch fd_abi_b32 plain; do python3 - $OUT/$f.json <<'PY'
import json,sys
lines=[l for l in open(sys.argv[1]).read().splitlines() if l.startswith('{')]
d=json.loads(lines[-1])
print(sys.argv[1].split('/')[-1], 'ms/step %.4f kern %.4f frac %.3f settle %s tune %s'%(d['ms_per_step'], d['roofline']['kernel_ms_avg'], d['roofline']['frac'], d['config']['settle'], d['config']['placement_tuning']))
PY
done
