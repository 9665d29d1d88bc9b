#!/usr/bin/env python3
"""Reads the kernel trace of tools/idle_transient.py: per burst, durations of the chain kernel in groups of 8 launches."""
import csv
import sys

rows = [r for r in csv.DictReader(open(sys.argv[1]))]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
idles = [0.0, 0.001, 0.003, 0.01, 0.03, 0.1, 0.3, 1.0, 3.0, 0.0]
bursts, cur = [], []
for r in rows:
    n = r["Kernel_Name"]
    if "cos_" in n and "elementwise" in n or "cos_kernel" in n:
        bursts.append(cur)
        cur = []
    elif "chain_kernel" in n and "chain_dyn" not in n:
        cur.append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
bursts = bursts[1:]      # the first segment is setup + settling
print("# chain kernel launch durations (us), mean of launches [0-8) [8-16) ... of each 160-launch burst")
for idle, b in zip(idles, bursts):
    g = [sum(b[i:i + 8]) / len(b[i:i + 8]) for i in range(0, len(b), 8)]
    print(f"idle {idle * 1e3:7.0f} ms | burst avg {sum(b) / len(b):6.1f} | first 24: {sum(b[:24]) / 24:6.1f} | " + " ".join(f"{v:4.0f}" for v in g))
