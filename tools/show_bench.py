#!/usr/bin/env python3
"""Pretty-print bench.py JSON lines from the files given as arguments, or from stdin."""
import fileinput, json
for l in fileinput.input():
    l = l.strip()
    if not l.startswith("{"):
        continue
    d = json.loads(l)
    r = d["roofline"]
    print("%.4f ms/step  %.3e samples/s  %s kern %.4f ms  %.0f %s frac %.3f  tune %s  %s" % (
        d["ms_per_step"], d["value"], r["kernel"], r["kernel_ms_avg"], r["achieved"], r["unit"], r["frac"],
        d["config"].get("placement_tuning"), [p for p in d["config"]["plan"] if "ring" in p]))
