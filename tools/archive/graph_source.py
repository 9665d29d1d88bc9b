#!/usr/bin/env python3
"""Print what the engine would compile for a saved graph (the reference's DSPConfig JSON): the plan (one kernel,
segments around FIR / Fuzz nodes, or run by run) and the generated translation unit of every graph kernel.
Needs no GPU.   usage: graph_source.py patch.json | graph_source.py test:<name in tests/graphs.py>"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from __graft_entry__ import load_package
E = load_package()
from dsp_stuff_amd import graph as G
arg = sys.argv[1] if len(sys.argv) > 1 else "test:diamond"
if arg.startswith("test:"):
    import graphs
    text = getattr(graphs, arg[5:])()
else:
    text = open(arg).read()
g = G.Graph(text)
plan = G.fused_plan(g)
if plan is not None:
    print("// one kernel for the whole graph (%d nodes)" % len(plan[0]))
    print(E.graph_source(*plan))
else:
    steps = G.series_plan(g)
    if steps is None:
        runs, _ = G.plan_runs(g)
        print("// run by run: %d chain engines: %s" % (len(runs), [[m.id for m in r.nodes] for r in runs]))
    else:
        for k, (kind, *what) in enumerate(steps):
            if kind == "graph":
                print("// segment %d: graph kernel, %d nodes" % (k, len(what[0])))
                print(E.graph_source(*what[:2]))
            else:
                print("// segment %d: %s node (its own kernel)\n" % (k, "FIR" if what[0].kind == E.FIR else "Fuzz"))
