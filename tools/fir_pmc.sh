#!/bin/bash
# rocprofv3 evidence for config 4 (262144 ch, 4096-tap FIR): kernel trace + PMC passes (each counter group in its own
# pass, counters only with --kernel-trace).  FETCH/WRITE are calibrated on fir_append_kernel of the same run: it reads
# N*B*4 bytes and writes N*B*4 bytes.  Summary: tools/fir_pmc_report.py <dir> <round>.  Extra environment (e.g.
# DSPFX_FIR_SKEW=0) is inherited by the benchmark.
set -u
R=${1:-r02}
OUT=/root/repo/gpurun_out/firpmc_$R; rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
B="python3 /root/repo/bench.py --config cfg4 --steps 20 --warmup 5 --no-cpu-baseline"
rocprofv3 -L > $OUT/counters.txt 2>&1
grep -o "SQ_[A-Z_0-9]*MFMA[A-Z_0-9]*" $OUT/counters.txt | sort -u > $OUT/mfma_counters.txt
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o cfg4 -- $B > $OUT/trace_bench.json 2>$OUT/trace.err
pass() { # name, counters...
  n=$1; shift
  rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d $OUT/pmc_$n -o cfg4 -- $B > $OUT/pmc_${n}_bench.json 2>$OUT/pmc_$n.err || echo "pass $n failed"
}
pass mfma SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE
pass mops SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_INSTS_MFMA SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY
pass valu SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM
pass fetch FETCH_SIZE
pass write WRITE_SIZE
find $OUT -name "*.csv" | head -40
