cd /tmp
for T in 64 256 512 1024 2048 4096; do
  for sp in 0 1; do
    DSPFX_FIR_SPLIT=$sp python3 /root/repo/bench.py --no-cpu-baseline --no-others --config cfg4 --taps $T --steps 50 --warmup 40 2>/dev/null | python3 -c "
import sys,json
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); r=d['roofline']; print('T %5d split %d  kernel %.4f ms  step %.4f ms  %s  algorithmic %.1f TFLOP/s'%($T,$sp,r['kernel_ms_avg'],d['ms_per_step'],r['kernel'],r.get('algorithmic_tflops', r['achieved'])))"
  done
done
