#!/usr/bin/env python3
"""Summarise a tools/clock_watch.sh log: shader clock and socket power over the busy phase (power > 900 W)."""
import re, sys
for fn in sys.argv[1:]:
    rows = []
    for l in open(fn):
        for card in l.split("|")[1:]:
            s = re.search(r"sclk\[\d+: (\d+)Mhz", card); p = re.search(r"power_uW=(\d+)", card)
            if s and p:
                rows.append((int(s.group(1)), int(p.group(1)) / 1e6))
    busy = [r for r in rows if r[1] > 900]
    if busy:
        print("%s: busy samples %d  sclk mean %.0f min %d MHz  power mean %.0f max %.0f W" % (
            fn, len(busy), sum(r[0] for r in busy) / len(busy), min(r[0] for r in busy),
            sum(r[1] for r in busy) / len(busy), max(r[1] for r in busy)))
    else:
        print(fn, ": no busy samples", len(rows))
