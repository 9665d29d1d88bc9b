#!/bin/bash
set -u
OUT=/root/repo/gpurun_out/r02c; mkdir -p $OUT
cd /root/repo
timeout 1700 python -m pytest tests -m gpu -q > $OUT/pytest.log 2>&1; echo "pytest rc $?"; tail -15 $OUT/pytest.log
cd /tmp && export TMPDIR=/tmp
python3 /root/repo/bench.py --config cfg4 --steps 50 --warmup 20 --no-cpu-baseline > $OUT/cfg4.json 2>$OUT/cfg4.err; tail -c 1500 $OUT/cfg4.json
DSPFX_FIR_FUSE=0 python3 /root/repo/bench.py --config cfg4 --steps 50 --warmup 20 --no-cpu-baseline > $OUT/cfg4_nofuse.json 2>$OUT/cfg4_nofuse.err; tail -c 600 $OUT/cfg4_nofuse.json
