#!/usr/bin/env python3
"""Per-launch durations of bench.py's timed regions from a rocprofv3 --kernel-trace CSV.

bench.py brackets every timed region with two one-workgroup marker kernels (torch sin_ before, cos_ after);
the launches between a marker pair are that config's K timed steps.  Prints / writes, per config in run order:
the dominant kernel, its per-launch durations (us), their average, the gaps between launches, and the same for
the `pre` launches of that kernel that ran immediately before the region (settling + warm-up tail).

    python tools/trace_phases.py <kernel_trace.csv> [bench_line.json] [out.json]
"""
import collections
import csv
import json
import statistics
import sys


def main():
    rows = [r for r in csv.DictReader(open(sys.argv[1]))]
    rows.sort(key=lambda r: int(r["Start_Timestamp"]))
    bench = None
    if len(sys.argv) > 2:
        try:
            bench = json.loads(open(sys.argv[2]).read().strip().splitlines()[-1])
        except Exception:
            bench = None
    is_sin = lambda n: "sin_kernel" in n or "sin_" in n and "asin" not in n and "elementwise" in n
    is_cos = lambda n: "cos_kernel" in n or "cos_" in n and "acos" not in n and "elementwise" in n
    regions, start = [], None
    for i, r in enumerate(rows):
        n = r["Kernel_Name"]
        if is_sin(n):
            start = i
        elif is_cos(n) and start is not None:
            regions.append((start, i))
            start = None
    out = {"trace": sys.argv[1], "regions": []}
    names = (bench or {}).get("timed_order") or ["cfg5", "cfg3", "cfg2", "cfg4"]
    headline = "cfg5"
    k = -1
    for (a, b) in regions:
        inner = rows[a + 1:b]
        if not inner:            # the markers' own warm-up at start-up
            continue
        k += 1
        cnt = collections.Counter(r["Kernel_Name"] for r in inner)
        tot = collections.Counter()
        for r in inner:
            tot[r["Kernel_Name"]] += int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
        dom = max(tot, key=tot.get)
        launches = [r for r in inner if r["Kernel_Name"] == dom]
        dur = [(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3 for r in launches]
        gaps = [(int(y["Start_Timestamp"]) - int(x["End_Timestamp"])) / 1e3 for x, y in zip(inner, inner[1:])]
        span = (int(inner[-1]["End_Timestamp"]) - int(inner[0]["Start_Timestamp"])) / 1e3
        pre = [r for r in rows[:a] if r["Kernel_Name"] == dom][-64:]
        pre_dur = [(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3 for r in pre]
        reg = {"config": names[k] if k < len(names) else str(k), "kernel": dom[:160], "launches": len(dur),
               "avg_us": sum(dur) / len(dur), "median_us": statistics.median(dur), "min_us": min(dur), "max_us": max(dur),
               "span_us_per_step": span / len(dur), "gap_median_us": statistics.median(gaps) if gaps else None,
               "other_kernels_in_region": {n[:80]: c for n, c in cnt.items() if n != dom},
               "durations_us": [round(d, 1) for d in dur],
               "preceding_64_launches_us": [round(d, 1) for d in pre_dur]}
        if bench is not None:
            b = bench if reg["config"] == headline else (bench.get("other_configs") or {}).get(reg["config"])
            if b and "roofline" in b:
                reg["bench_reported"] = {"kernel_ms_avg": b["roofline"]["kernel_ms_avg"], "ms_per_step": b["ms_per_step"],
                                         "frac": b["roofline"]["frac"]}
                reg["trace_over_bench"] = reg["avg_us"] / (b["roofline"]["kernel_ms_avg"] * 1e3)
        out["regions"].append(reg)
        print(f'{reg["config"]}: {reg["kernel"][:60]}  n={len(dur)} avg {reg["avg_us"]:.1f} us  median {reg["median_us"]:.1f}  '
              f'span/step {reg["span_us_per_step"]:.1f}  bench {reg.get("bench_reported")}')
    if len(sys.argv) > 3:
        json.dump(out, open(sys.argv[3], "w"), indent=1)


if __name__ == "__main__":
    main()
