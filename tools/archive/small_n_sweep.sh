for v in "f=8,cpl=2" "f=8,cpl=1" "f=16,cpl=1" "f=16,cpl=2" "f=32,cpl=1"; do
  for t in 256 0; do
    echo "== variant $v tile $t"
    DSPFX_VARIANT="$v" python bench.py --config cfg2 --no-cpu-baseline --tile $t --probe 1 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print(d['roofline']['kernel'], 'ms/step', round(d['ms_per_step'],4), 'kernel_ms', round(d['roofline']['kernel_ms_avg'],4), 'GB/s', round(d['roofline']['achieved']))"
  done
done
