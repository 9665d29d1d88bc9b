"""Reference-semantics evaluation of a whole saved graph (DAG) on the CPU oracle.

TEST INFRASTRUCTURE ONLY -- PARITY UNPINNED (see dspfx_oracle.h).  Every node is evaluated the way the
reference's `Perform` wrapper does it (dsp-stuff/src/node.rs:267-352): per 128-frame block, every input port
is `collect_and_average` over its connected pipes in link order (zeros when none), then `process`.
`graph` is a parsed `dsp_stuff_amd.graph.Graph` (only its structure and node descriptors are used).
"""
from __future__ import annotations

import ctypes as C

import numpy as np

import oracle as O


def _avg(L, srcs, nf):
    out = np.zeros(nf, np.float32)                      # node.rs:288: the port buffer starts zeroed
    arr = (C.POINTER(C.c_float) * max(1, len(srcs)))()
    keep = [np.ascontiguousarray(s, np.float32) for s in srcs]
    for i, s in enumerate(keep):
        arr[i] = s.ctypes.data_as(C.POINTER(C.c_float))
    L.orc_collect_and_average(out.ctypes.data_as(C.POINTER(C.c_float)), arr, len(keep), nf)
    return out


def run_graph(graph, x, block=O.BUF_SIZE):
    """x: [n_frames][n_channels] (the Input node's signal) -> the Output node's signal, same shape."""
    x = np.asarray(x, np.float32)
    nf_total, N = x.shape
    L = O.lib()
    out = np.empty_like(x)
    for c in range(N):
        nodes = {i: O.node_from_desc(n.spec.oracle_desc()) for i, n in graph.nodes.items() if n.spec is not None}
        for f0 in range(0, nf_total, block):
            nf = min(block, nf_total - f0)
            val = {-1: np.zeros(nf, np.float32)}       # graph.ZERO: the unselected output of a demux
            for nid in graph.order:
                n = graph.nodes[nid]
                if n.typename == "input":
                    val[nid] = x[f0:f0 + nf, c]
                    continue
                a = _avg(L, [val[s] for s in n.main], nf)
                if n.typename == "output":
                    out[f0:f0 + nf, c] = a
                    continue
                b = _avg(L, [val[s] for s in n.side], nf) if n.side else None
                ctl = None
                if n.ctl:
                    ctl = [None, None, None]
                    for k, srcs in n.ctl.items():
                        ctl[k] = _avg(L, [val[s] for s in srcs], nf)
                val[nid] = nodes[nid].process(a, b, ctl)
    return out
