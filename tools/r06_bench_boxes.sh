#!/bin/bash
# One more fresh box: the driver's command, one summary line appended to gpurun_out/r06_bench_boxes.txt (run it once per gpurun call).
python3 bench.py --gpus 1 --steps 20 --warmup 5 2>/dev/null | python3 -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1]); r = d['roofline']; o = d['other_configs']
f = lambda v: '%s k %.4f s %.4f frac %.3f by_step %.3f' % (v['roofline']['kernel'], v['roofline']['kernel_ms_avg'], v['ms_per_step'], v['roofline']['frac'], v['roofline']['frac_by_step'])
print('cfg5 k %.4f s %.4f frac %.3f by_step %.3f fixed %.0f us | ' % (r['kernel_ms_avg'], d['ms_per_step'], r['frac'], r['frac_by_step'], d['region_fixed_cost_us']) + ' | '.join(k + ' ' + f(v) for k, v in o.items()) +
      ' | paced p50 %.3f p99 %.3f misses %d | host 1M %.2f ms, rt %d ch p99 %.2f ms' % (d['paced']['latency_ms']['p50'], d['paced']['latency_ms']['p99'], d['paced']['deadline_misses'],
      d['host_path']['cfg5_shard']['ms_per_block_p50'], d['host_path']['largest_realtime_pow2']['channels'], d['host_path']['largest_realtime_pow2']['ms_per_block_p99']))" >> gpurun_out/r06_bench_boxes.txt
tail -1 gpurun_out/r06_bench_boxes.txt
