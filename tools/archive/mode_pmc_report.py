#!/usr/bin/env python3
"""Split the chain-kernel dispatches of a rocprofv3 --pmc run into fast / slow by duration and compare counters."""
import csv, sys, collections, statistics
rows = [r for r in csv.DictReader(open(sys.argv[1])) if "chain_kernel" in r["Kernel_Name"]]
by = collections.defaultdict(dict)
for r in rows:
    d = by[int(r["Dispatch_Id"])]
    d["dur"] = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
    d[r["Counter_Name"]] = float(r["Counter_Value"])
ds = [by[k] for k in sorted(by)]
durs = sorted(d["dur"] for d in ds)
thr = (durs[len(durs) // 10] + durs[-len(durs) // 10]) / 2
fast = [d for d in ds if d["dur"] < thr]; slow = [d for d in ds if d["dur"] >= thr]
print("dispatches %d: fast %d (median %.1f us)  slow %d (median %.1f us)  threshold %.1f" % (
    len(ds), len(fast), statistics.median(d["dur"] for d in fast) if fast else 0, len(slow),
    statistics.median(d["dur"] for d in slow) if slow else 0, thr))
for c in sorted(k for k in ds[0] if k != "dur"):
    f = statistics.mean(d[c] for d in fast) if fast else 0; s = statistics.mean(d[c] for d in slow) if slow else 0
    print("  %-44s fast %.4g   slow %.4g   slow/fast %.3f" % (c, f, s, s / f if f else float("nan")))
