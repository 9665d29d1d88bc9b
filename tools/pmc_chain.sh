#!/bin/bash
# Counter evidence for the chain kernels that bench.py times: cfg5, cfg3, cfg2.  Each counter in its own pass, counters only
# with --kernel-trace (MI355X_MICROARCH.md: FETCH_SIZE and WRITE_SIZE do not fit one pass).  Calibrated on the empty-chain kernel of
# identical access width in the SAME pass (tools/pmc_chain_workload.py).  Report: tools/pmc_chain_report.py -> profiles/<round>_pmc_<cfg>.json (round = $DSPFX_ROUND)
# and profiles/traffic.json (every entry from this run).
set -u
OUT=/root/repo/gpurun_out/pmc_chain; rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for cfg in cfg5 cfg3 cfg2; do
  for c in FETCH_SIZE WRITE_SIZE; do
    rocprofv3 --kernel-trace --pmc $c --output-format csv -d $OUT/${cfg}_$c -o p -- python3 /root/repo/tools/pmc_chain_workload.py $cfg 40 > $OUT/${cfg}_$c.log 2>$OUT/${cfg}_$c.err || echo "pass $cfg $c failed"
  done
done
python3 /root/repo/tools/pmc_chain_report.py $OUT $OUT/profiles
