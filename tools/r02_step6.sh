#!/bin/bash
set -u
OUT=/root/repo/gpurun_out/r02f; mkdir -p $OUT
cd /root/repo
timeout 1700 python -m pytest tests -m gpu -q > $OUT/pytest.log 2>&1; echo "pytest rc $?"; tail -6 $OUT/pytest.log
cd /tmp && export TMPDIR=/tmp
B="python3 /root/repo/bench.py --no-cpu-baseline"
for m in abi torch; do
  DSPFX_BENCH_COMM=$m DSPFX_BENCH_FORCE_DIST=1 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/fd_$m -o fd -- $B --steps 64 --warmup 16 --no-others > $OUT/fd_$m.json 2>$OUT/fd_$m.err
done
bash /root/repo/tools/fir_pmc.sh r02 > $OUT/fir_pmc.log 2>&1; tail -5 $OUT/fir_pmc.log
