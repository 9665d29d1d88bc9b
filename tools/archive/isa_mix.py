#!/usr/bin/env python3
"""Instruction mix of each kernel's hottest (largest) basic block in a gfx950 .s file."""
import collections
import re
import sys

src = open(sys.argv[1]).read()
pat = sys.argv[2] if len(sys.argv) > 2 else ""
for m in re.finditer(r'^(_Z\w+):[^\n]*\n(.*?)\n\s*s_endpgm', src, re.S | re.M):
    name, body = m.group(1), m.group(2)
    if pat and pat not in name:
        continue
    blocks = re.split(r'\n\.LBB[0-9_]+:[^\n]*', body)
    big = max(blocks, key=lambda b: sum(1 for l in b.split('\n') if l.strip() and not l.strip().startswith((';', '.'))))
    ops = collections.Counter(l.split()[0] for l in big.split('\n') if l.strip() and not l.strip().startswith((';', '.')))
    groups = collections.Counter()
    for k, v in ops.items():
        g = ('valu' if k.startswith('v_') else 'salu' if k.startswith('s_') else
             'vmem' if k.startswith(('global_', 'buffer_', 'flat_', 'scratch_')) else 'lds' if k.startswith('ds_') else 'other')
        groups[g] += v
    print(name[:110])
    print("  hottest block:", sum(ops.values()), dict(groups))
    print("  ", ops.most_common(int(sys.argv[3]) if len(sys.argv) > 3 else 30))
