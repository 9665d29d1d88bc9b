#!/bin/bash
set -u
OUT=/root/repo/gpurun_out/r02d; mkdir -p $OUT
cd /root/repo
timeout 1700 python -m pytest tests -m gpu -q > $OUT/pytest.log 2>&1; echo "pytest rc $?"; tail -8 $OUT/pytest.log
cd /tmp && export TMPDIR=/tmp
python3 /root/repo/bench.py --config cfg4 --steps 50 --warmup 20 --no-cpu-baseline > $OUT/cfg4.json 2>$OUT/cfg4.err; tail -c 700 $OUT/cfg4.json | head -c 600; echo
DSPFX_FIR_FUSE=0 python3 /root/repo/bench.py --config cfg4 --steps 50 --warmup 20 --no-cpu-baseline > $OUT/cfg4_nofuse.json 2>$OUT/cfg4_nofuse.err; tail -c 700 $OUT/cfg4_nofuse.json | head -c 600; echo
DSPFX_BENCH_FORCE_DIST=1 python3 /root/repo/bench.py --steps 50 --warmup 20 --no-cpu-baseline --no-others > $OUT/cfg5_forcedist.json 2>$OUT/cfg5_forcedist.err; tail -c 300 $OUT/cfg5_forcedist.err; tail -c 2500 $OUT/cfg5_forcedist.json | head -c 900
