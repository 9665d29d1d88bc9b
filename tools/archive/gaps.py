#!/usr/bin/env python3
"""Idle gaps between consecutive chain kernels in a rocprofv3 kernel-trace CSV."""
import csv, sys, statistics
rows = [r for r in csv.DictReader(open(sys.argv[1]))]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
ch = [r for r in rows if "chain_kernel" in r["Kernel_Name"]][-150:]
gaps = [(int(b["Start_Timestamp"]) - int(a["End_Timestamp"])) / 1e3 for a, b in zip(ch, ch[1:])]
dur = [(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3 for r in ch]
print("chain kernels: dur median %.1f us; gap median %.2f us, p90 %.2f, max %.2f" % (statistics.median(dur), statistics.median(gaps), sorted(gaps)[int(.9*len(gaps))], max(gaps)))
others = {}
for r in rows[-600:]:
    if "chain_kernel" in r["Kernel_Name"]: continue
    others.setdefault(r["Kernel_Name"][:40], []).append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
for k, v in others.items(): print("  %-40s n=%d median %.2f us" % (k, len(v), statistics.median(v)))
