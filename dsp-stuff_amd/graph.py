"""Whole saved graphs (DAGs), not just linear chains: the reference's `DSPConfig` with fan-out, fan-in
(`collect_and_average` over several pipes, dsp-stuff/src/node.rs:162-194,267-352), Add / Mix fed by two
different branches, control ports fed by other nodes, generator sources, several links into the Output node.

A graph of at most 16 fusable nodes becomes ONE kernel generated for its wiring (`dspfx_graph_set`, csrc/graph_kernel.hip.h):
node outputs stay in registers and a block costs one read of the Input node's buffer and one write of the Output node's.

A graph whose FIR / Fuzz nodes (kernels of their own) are passed by all the signal -- an amp chain into a cabinet
impulse response into a reverb -- is cut at those nodes, a graph of more than 16 nodes where a single signal crosses,
and every segment is one such kernel (`segment_plan`).

Anything else (no such cutting, a link that bypasses a FIR or Fuzz node, a channel count that is not a multiple of 64, no
run-time compiler) is cut into maximal linear runs; each run is one fused `Engine` (one kernel launch per block, its
own per-channel state), runs are evaluated in topological order, and the only extra device work is
`dspfx_link_average` where a port has more than one incoming link.  What a run consumes:

  main port   1 link  -> the producer's buffer, hop applied by the engine (LINK_INPUT)
              k links -> dspfx_link_average into a scratch buffer, taken raw
              0 links -> zeros (node.rs:288: the port's buffer stays zeroed)
  "b" port    same three cases through the engine's side input (LINK_SIDE_RAW for k > 1)
  slider port 1 link  -> control port of dspfx_process_ctl; more than one link is rejected

Mux / Demux (nodes/mux.rs:42-55, nodes/demux.rs:42-58) are pure routing and cost no kernel of their own: a mux is
the identity on its selected port (links into the other port are computed but unused, as in the reference), a
demux is the identity whose unselected output port delivers zeros -- still a connected pipe for whoever averages it.

The display-only nodes (pitch detector, wave view, spectrogram) produce no signal; they are dropped on import.

Not expressible: cycles (the reference's scheduler would deadlock on them too), muff (GPL crate, source absent),
slider ports with fan-in.
"""
from __future__ import annotations

import json
from typing import Dict, List, Optional, Tuple

from . import (ADD, DISTORT, ERR_UNSUPPORTED, FIR, FUZZ, GAIN, GRAPH_INPUT, GRAPH_INPUT2, GRAPH_INPUTS, GRAPH_MAX_IO, GRAPH_MAX_NODES, GRAPH_ZERO,
               LINK_INPUT, LINK_INTERNAL, LINK_SIDE_RAW, MIX, PORT_MAIN, PORT_RAW, PORT_SIDE, PORT_SLIDER, SIGNAL_GEN, DspfxError, Engine,
               NodeSpec)
from .config import _TABLE, DspConfigError, _node_from_cfg

_UNSUPPORTED = {"muff"}                                # GPL crate, source not in the reference tree
_SINKS = {"pitch", "wave_view", "spectrogram"}          # display-only: `process` writes no output (pitch.rs:120-146,
                                                        # wave_view.rs:157-175), so they and the links into them are dropped
_ROUTING = {"mux", "demux"}
ZERO = -1          # pseudo producer: the unselected output port of a demux (a connected pipe that carries zeros)


class _GNode:
    __slots__ = ("id", "typename", "cfg", "spec", "main", "side", "ctl", "outs")

    def __init__(self, nid, typename, cfg):
        self.id, self.typename, self.cfg = nid, typename, cfg
        self.spec = None
        self.main: List[int] = []          # producer node ids, link order
        self.side: List[int] = []
        self.ctl: Dict[int, List[int]] = {}   # slider index -> producers
        self.outs: List[Tuple[int, str]] = []  # (consumer id, port name)


class Graph:
    """Parsed DSPConfig: nodes with their incoming links per port, in document order."""

    def __init__(self, text: str, page_round: bool = False):
        try:
            doc = json.loads(text)
            raw_nodes = doc["nodes"]
            raw_links = [(tuple(map(int, l["lhs"])), tuple(map(int, l["rhs"]))) for l in doc["links"]]
        except (KeyError, TypeError, ValueError) as e:
            raise DspConfigError(f"not a DSPConfig document: {e}") from None
        self.nodes: Dict[int, _GNode] = {}
        self.dropped: List[int] = []
        for n in raw_nodes:
            tn = n["typename"]
            if tn in _SINKS:
                self.dropped.append(int(n["id"]))
                continue
            if tn in _UNSUPPORTED:
                raise DspConfigError(f"node type {tn!r} is outside the accelerated path")
            if tn not in _TABLE and tn not in _ROUTING and tn not in ("input", "output"):
                raise DspConfigError(f"unknown node type {tn!r}")
            g = _GNode(int(n["id"]), tn, n["cfg"])
            if tn in _TABLE:
                g.spec = _node_from_cfg(tn, n["cfg"], page_round)
            elif tn in _ROUTING:
                sel = n["cfg"].get("in_port" if tn == "mux" else "out_port", "A")
                if sel not in ("A", "B"):
                    raise DspConfigError(f"{tn} node {g.id}: unknown port selection {sel!r}")
                g.spec = NodeSpec(GAIN, [1.0])          # x * 1.0f: the copy_from_slice of mux.rs:54 / demux.rs:51,54
            self.nodes[g.id] = g
        self.inputs = [i for i, n in self.nodes.items() if n.typename == "input"]
        self.outputs = [i for i, n in self.nodes.items() if n.typename == "output"]
        if len(self.inputs) > 1 or len(self.outputs) != 1:
            raise DspConfigError("expected at most one input node and exactly one output node")

        def port_name(node_id, port_id, which):
            for name, pid in self.nodes[node_id].cfg.get(which, {}).items():
                if int(pid) == port_id:
                    return name
            raise DspConfigError(f"link refers to unknown {which[:-1]} port {port_id} of node {node_id}")

        for (ln, lp), (rn, rp) in raw_links:
            if rn in self.dropped:
                continue
            if ln not in self.nodes or rn not in self.nodes:
                raise DspConfigError("link refers to a missing node")
            oname = port_name(ln, lp, "outputs")
            pname = port_name(rn, rp, "inputs")
            dst = self.nodes[rn]
            src = self.nodes[ln]
            if src.typename == "demux" and oname != src.cfg.get("out_port", "A").lower():
                ln = ZERO                               # demux.rs:49-56: the other output keeps its zeroed buffer
            else:
                src.outs.append((rn, pname))
            if dst.typename == "output":
                dst.main.append(ln)
                continue
            if dst.typename == "mux":
                if pname == dst.cfg.get("in_port", "A").lower():
                    dst.main.append(ln)
                elif ln != ZERO:
                    src.outs.pop()                      # mux.rs:47-50: the unselected port is never read
                    src.outs.append((rn, "unused"))     # ... but it is still a consumer: the producer ends its run
                continue
            if dst.typename == "demux":
                dst.main.append(ln)
                continue
            _, fields, main_port, ctl_ports = _TABLE[dst.typename]
            if pname == main_port:
                dst.main.append(ln)
            elif pname == "b" and dst.spec.kind in (ADD, MIX):
                dst.side.append(ln)
            elif pname in ctl_ports:
                dst.ctl.setdefault(fields.index(pname), []).append(ln)
            else:
                raise DspConfigError(f"node {rn} ({dst.typename}) has no input port {pname!r}")
        for n in self.nodes.values():
            for k, src in n.ctl.items():
                if len(src) > 1:
                    raise DspConfigError(f"slider port {k} of node {n.id} averages several links")
        self.order = self._toposort()

    def producers(self, n: _GNode) -> List[int]:
        return [s for s in n.main + n.side + [s for v in n.ctl.values() for s in v] if s != ZERO]

    def _toposort(self) -> List[int]:
        indeg = {i: len(self.producers(n)) for i, n in self.nodes.items()}
        ready = [i for i in self.nodes if indeg[i] == 0]          # document order: deterministic
        order = []
        while ready:
            i = ready.pop(0)
            order.append(i)
            for j, port in self.nodes[i].outs:
                if port == "unused":
                    continue
                indeg[j] -= 1
                if indeg[j] == 0:
                    ready.append(j)
        if len(order) != len(self.nodes):
            raise DspConfigError("graph has a cycle")
        return order


class _Run:
    """A maximal linear run of nodes = one fused engine."""

    def __init__(self, first: _GNode):
        self.nodes: List[_GNode] = [first]
        self.engine: Optional[Engine] = None
        self.out = None
        self.scratch_main = None
        self.scratch_side = None


def plan_runs(g: Graph):
    """Cut the graph into maximal linear runs (each becomes one fused engine).  A node joins its producer's run
    when it is the producer's only consumer and is fed by nothing else on its main port."""
    run_of: Dict[int, _Run] = {}
    runs: List[_Run] = []
    for nid in g.order:
        n = g.nodes[nid]
        if n.spec is None:
            continue
        prev = g.nodes[n.main[0]] if len(n.main) == 1 and n.main[0] != ZERO else None
        joinable = (prev is not None and prev.spec is not None and len(prev.outs) == 1
                    and n.spec.kind != SIGNAL_GEN)
        if joinable:
            r = run_of[prev.id]
            # one side input per engine, shared by every Add/Mix of its chain: two of them may share a run
            # only when neither has its "b" port connected (both read zeros)
            mixers = [m for m in r.nodes if m.spec.kind in (ADD, MIX)]
            clash = n.spec.kind in (ADD, MIX) and mixers and (n.side or any(m.side for m in mixers))
            if len(r.nodes) < 32 and not clash:
                r.nodes.append(n)
                run_of[nid] = r
                continue
        r = _Run(n)
        runs.append(r)
        run_of[nid] = r
    return runs, run_of


def run_link_flags(r: _Run) -> int:
    head = r.nodes[0]
    side_node = next((m for m in r.nodes if m.side), None)
    flags = LINK_INTERNAL
    if len(head.main) == 1 and head.spec.kind != SIGNAL_GEN:
        flags |= LINK_INPUT
    if side_node is not None and len(side_node.side) > 1:
        flags |= LINK_SIDE_RAW
    return flags


def fused_plan(g: Graph):
    """The graph as `dspfx_graph_set` takes it: (node specs in topological order, links), or None when it cannot
    be one kernel (too many nodes, a FIR or Fuzz node)."""
    order = [nid for nid in g.order if g.nodes[nid].spec is not None]
    if len(order) > GRAPH_MAX_NODES:
        return None
    for nid in order:
        sp = g.nodes[nid].spec
        if sp.kind == FIR or (sp.kind == DISTORT and sp.mode == FUZZ):
            return None
    idx = {nid: i for i, nid in enumerate(order)}

    def src(s):
        if s == ZERO:
            return GRAPH_ZERO
        return GRAPH_INPUT if g.nodes[s].typename == "input" else idx[s]

    links = []
    for nid in order:
        n = g.nodes[nid]
        links += [(src(s), idx[nid], PORT_MAIN) for s in n.main]
        links += [(src(s), idx[nid], PORT_SIDE) for s in n.side]
        for k, srcs in sorted(n.ctl.items()):
            links += [(src(s), idx[nid], PORT_SLIDER + k) for s in srcs]
    links += [(src(s), len(order), PORT_MAIN) for s in g.nodes[g.outputs[0]].main]
    return [g.nodes[nid].spec for nid in order], links


def _unfusable(sp) -> bool:
    return sp.kind == FIR or (sp.kind == DISTORT and sp.mode == FUZZ)


def segment_plan(g: Graph, max_nodes: int = GRAPH_MAX_NODES):
    """A graph too large for one kernel, or with FIR / Fuzz nodes (kernels of their own), as a SERIES of engines:
    [("graph", specs, links, in_ref, in2_ref), ("node", spec, in_ref), ("graph", ...), ...] -- a ref is the index of the
    step whose output block is read, -1 = the graph's Input block.  Two kinds of boundary:

      * in front of a FIR / Fuzz node that all the live signal goes into -- an amp chain into a cabinet impulse
        response into a reverb: the segment's Output is that node's averaged main port, the node follows, the next
        segment's Input is the node's output;
      * where a segment would exceed `max_nodes`, at a point of the evaluation order that ONE new signal crosses: the
        segment's Output link hands that signal over as it is (PORT_RAW), the next segment's Input is that buffer.

    At either boundary one OLDER signal -- the current segment's own Input or second block, e.g. the dry signal of a
    wet / dry rig -- may stay alive beside the new one: the next segment reads it as its second block (GRAPH_INPUT2).

    A FIR / Fuzz node fed by ONE signal that also goes on beside it (the dry path around a cabinet) is allowed: the
    segment hands that signal over as it is, the node ("node_hop" step) applies the hop itself, and the following
    segment reads the node as its Input and the signal as its second block (GRAPH_INPUT2, the engine's `side`).

    Every segment is a sub-DAG with one Output for `dspfx_graph_set`.  None when no such cutting exists
    (some link bypasses a FIR / Fuzz node, one of them has a connected slider port, a stretch of more than
    `max_nodes` nodes has no single-signal crossing)."""
    nodes = [nid for nid in g.order if g.nodes[nid].spec is not None]
    out_id = g.outputs[0]
    in_id = g.inputs[0] if g.inputs else None
    fed = {}
    for nid in g.order:
        fed[nid] = g.nodes[nid].typename == "input" or any(fed[p] for p in g.producers(g.nodes[nid]))
    # sources fed by nothing (generators, unplugged effects) are evaluated as late as possible: right before their
    # first consumer, so that they do not keep a signal alive across the points where the graph could be cut
    order, placed = [], set()

    def place(nid):
        if nid in placed:
            return
        for p in g.producers(g.nodes[nid]):
            if g.nodes[p].spec is not None and not fed[p]:
                place(p)
        placed.add(nid)
        order.append(nid)

    for nid in nodes:
        if fed[nid]:
            place(nid)
    for nid in nodes:
        place(nid)
    pos = {nid: i for i, nid in enumerate(order)}
    pos[out_id] = len(order)
    users: Dict[int, List[int]] = {}                   # value (node id / input id) -> positions that read it
    for nid in order + [out_id]:
        for p in g.producers(g.nodes[nid]):
            users.setdefault(p, []).append(pos[nid])

    def live_after(values, p):
        """those of `values` still read at position p or later"""
        return [v for v in values if any(u >= p for u in users.get(v, []))]

    steps = []
    start = 0
    cur_in, ref_in = in_id, -1                         # the signal the current segment reads as its Input, and whose buffer it is
    cur_in2, ref_in2 = None, None                      # a second, older signal still alive (GRAPH_INPUT2)

    def emit(lo, hi, sink_links, raw):
        seg = order[lo:hi]
        idx = {nid: i for i, nid in enumerate(seg)}

        def src(sv):
            if sv == ZERO:
                return GRAPH_ZERO
            if sv == cur_in:
                return GRAPH_INPUT
            if sv == cur_in2:
                return GRAPH_INPUT2
            return idx[sv]                             # KeyError = a signal from further back: caught below

        links = []
        for nid in seg:
            n = g.nodes[nid]
            links += [(src(sv), idx[nid], PORT_MAIN) for sv in n.main]
            links += [(src(sv), idx[nid], PORT_SIDE) for sv in n.side]
            for kk, srcs in sorted(n.ctl.items()):
                links += [(src(sv), idx[nid], PORT_SLIDER + kk) for sv in srcs]
        links += [(src(sv), len(seg), PORT_MAIN | (PORT_RAW if raw else 0)) for sv in sink_links]
        steps.append(("graph", [g.nodes[nid].spec for nid in seg], links, ref_in, ref_in2))
        return len(steps) - 1

    def ref_of(sv):
        return ref_in if sv == cur_in else ref_in2

    try:
        i = start
        while i <= len(order):
            at_end = i == len(order)
            cut_node = (not at_end) and _unfusable(g.nodes[order[i]].spec)
            if at_end or cut_node:
                # the stretch [start, i) must fit: cut it where ONE new signal crosses (an older one may go on beside it)
                while i - start > max_nodes:
                    best = None
                    for p in range(start + 1, min(start + max_nodes, i - 1) + 1):
                        live = live_after([cur_in, cur_in2] + order[start:p], p)
                        olds = [v for v in live if v in (cur_in, cur_in2)]
                        news = [v for v in live if v not in olds]
                        if len(news) == 1 and len(olds) <= 1:
                            best = (p, news[0], olds)
                    if best is None:
                        return None
                    p, v, olds = best
                    k = emit(start, p, [v], True)
                    cur_in2, ref_in2 = (olds[0], ref_of(olds[0])) if olds else (None, None)
                    start, cur_in, ref_in = p, v, k
                if at_end:
                    emit(start, i, g.nodes[out_id].main, False)
                    break
                u = g.nodes[order[i]]
                if u.ctl or u.side:
                    return None
                live = live_after([cur_in, cur_in2] + order[start:i], i)
                carried = [v for v in live if v in (cur_in, cur_in2) and v not in u.main]   # older signals going around the node
                rest = [v for v in live if v not in carried]
                if len(carried) <= 1 and all(v in u.main and not any(x > i for x in users.get(v, [])) for v in rest):
                    # everything else alive here is read by this node's main port and by nothing later
                    keep = (carried[0], ref_of(carried[0])) if carried else (None, None)
                    k = emit(start, i, u.main, False)
                    steps.append(("node", u.spec, k))
                    cur_in2, ref_in2 = keep
                elif len(live) == 1 and u.main == live and (live[0] in order[start:i] or (live[0] == cur_in and start == i)):
                    # one signal feeds the node AND goes on beside it (the dry path around a cabinet): hand it over as it
                    # is, the node applies its own hop, the next segment reads the node as Input and the signal as Input 2
                    k = emit(start, i, live, True) if start < i else ref_in
                    steps.append(("node_hop", u.spec, k))
                    cur_in2, ref_in2 = live[0], k
                else:
                    return None
                start, cur_in, ref_in = i + 1, order[i], len(steps) - 1
            i += 1
    except KeyError:
        return None                                    # a link from before the previous boundary: no series form
    return steps


def _late_source_order(g: Graph) -> List[int]:
    """Topological order of the effect nodes in which sources fed by nothing (generators, unplugged effects) come as late
    as possible -- right before their first consumer -- so that they do not keep a signal alive across cutting points."""
    nodes = [nid for nid in g.order if g.nodes[nid].spec is not None]
    fed = {}
    for nid in g.order:
        fed[nid] = g.nodes[nid].typename == "input" or any(fed[p] for p in g.producers(g.nodes[nid]))
    order, placed = [], set()

    def place(nid):
        if nid in placed:
            return
        for p in g.producers(g.nodes[nid]):
            if g.nodes[p].spec is not None and not fed[p]:
                place(p)
        placed.add(nid)
        order.append(nid)

    for nid in nodes:
        if fed[nid]:
            place(nid)
    for nid in nodes:
        place(nid)
    return order


def region_plan(g: Graph, max_nodes: int = GRAPH_MAX_NODES, max_io: int = GRAPH_MAX_IO):
    """ANY graph as a short series of generated kernels: the evaluation order is cut into REGIONS of at most `max_nodes`
    fusable nodes, each one kernel with up to `max_io` input blocks (signals from the Input node, from earlier regions,
    from FIR / Fuzz nodes -- read as main inputs, Add / Mix side inputs or control signals alike) and up to `max_io`
    output blocks (every signal a later step still reads, handed over untouched).  FIR / Fuzz nodes keep their own
    kernels between the regions; the last region also evaluates the Output node's port.  Steps:

      ("region", specs, links, in_refs, n_out)     links as for dspfx_graph_set: sources GRAPH_INPUTS[k] = in_refs[k],
                                                    dst == len(specs) + m = output block m (block 0 of the last region
                                                    is the Output node's average, every other block a PORT_RAW handover)
      ("node", spec, main_refs, ctl_refs)          a FIR / Fuzz node: main port averaged over main_refs, slider k fed
                                                    by ctl_refs[k]
      ("output", refs)                             the Output node's average when no region is left to carry it

    A ref is -1 (the graph's Input block), (step index, output block) or None (a connected pipe of zeros).
    None when some region would need more than `max_io` blocks either way."""
    order = _late_source_order(g)
    out_id = g.outputs[0]
    pos = {nid: i for i, nid in enumerate(order)}
    pos[out_id] = len(order)
    last_use: Dict[int, int] = {}
    for nid in order + [out_id]:
        for p in g.producers(g.nodes[nid]):
            last_use[p] = max(last_use.get(p, -1), pos[nid])
    loc: Dict[int, object] = {}
    if g.inputs:
        loc[g.inputs[0]] = -1
    steps = []

    def ref(v):
        return None if v == ZERO else loc[v]

    i = 0
    while i < len(order):
        nid = order[i]
        n = g.nodes[nid]
        if _unfusable(n.spec):
            if any(len(srcs) != 1 for srcs in n.ctl.values()):
                return None        # a slider port averaging several links (the importer rejects those): never take just the first
            steps.append(("node", n.spec, [ref(v) for v in n.main], {k: ref(srcs[0]) for k, srcs in n.ctl.items()}))
            loc[nid] = (len(steps) - 1, 0)
            i += 1
            continue
        best = None
        end = i
        while end < len(order) and end - i < max_nodes and not _unfusable(g.nodes[order[end]].spec):
            end += 1
            region = order[i:end]
            inside = set(region)
            final = end == len(order)
            ext: List[int] = []
            consumers = region + ([out_id] if final else [])
            for c in consumers:
                for v in g.producers(g.nodes[c]):
                    if v not in inside and v not in ext:
                        ext.append(v)
            outs = [] if final else [v for v in region if last_use.get(v, -1) >= end]      # read by a later step
            n_out = (1 if final else 0) + len(outs)
            if len(ext) <= max_io and n_out <= max_io:
                best = (end, region, ext, outs, final)
        if best is None:
            return None
        end, region, ext, outs, final = best
        idx = {v: k for k, v in enumerate(region)}
        blk = {v: k for k, v in enumerate(ext)}

        def src(v):
            if v == ZERO:
                return GRAPH_ZERO
            return idx[v] if v in idx else GRAPH_INPUTS[blk[v]]

        links = []
        for v in region:
            m = g.nodes[v]
            links += [(src(s), idx[v], PORT_MAIN) for s in m.main]
            links += [(src(s), idx[v], PORT_SIDE) for s in m.side]
            for kk, srcs in sorted(m.ctl.items()):
                links += [(src(s), idx[v], PORT_SLIDER + kk) for s in srcs]
        nn = len(region)
        if final:
            links += [(src(s), nn, PORT_MAIN) for s in g.nodes[out_id].main]
        if not final and not outs:
            outs = [region[-1]]                          # nothing of this region is read later (dead branch): still one block
        base = 1 if final else 0
        for m_, v in enumerate(outs):
            links.append((idx[v], nn + base + m_, PORT_MAIN | PORT_RAW))
        steps.append(("region", [g.nodes[v].spec for v in region], links, [loc[v] for v in ext], base + len(outs)))
        for m_, v in enumerate(outs):
            loc[v] = (len(steps) - 1, base + m_)
        i = end
    if not steps or steps[-1][0] != "region":        # the graph ends in a FIR / Fuzz node (or has no effect node at all)
        steps.append(("output", [ref(v) for v in g.nodes[out_id].main]))
    return steps


def series_plan(g: Graph):
    """`segment_plan` for graphs that need it: None for a graph that is one kernel anyway."""
    return None if fused_plan(g) is not None else segment_plan(g)


class GraphEngine:
    """N independent copies of a saved graph.  `process(x)` takes the Input node's block [n_frames][N] (device
    tensor, the engine's layout) and returns the Output node's block.
    fused: None = one generated kernel for the whole graph when it can be had, else a series of them (segment_plan),
    else run by run; True = insist on the one kernel; False = always run by run."""

    def __init__(self, text: str, channels: int, max_frames: int = 128, device: int = 0, tile_channels: int = 0,
                 page_round: bool = False, fused: Optional[bool] = None, max_nodes: Optional[int] = None,
                 regions: Optional[bool] = None):
        import torch
        self.torch = torch
        self.g = Graph(text, page_round)
        self.N, self.B, self.tile = channels, max_frames, tile_channels
        self.dev = torch.device("cuda", device)
        self.fused: Optional[Engine] = None
        self.series, self.series_kind = [], []
        self.regions = []
        self.runs, self.run_of = [], {}
        self.zeros = torch.zeros(max_frames * channels, dtype=torch.float32, device=self.dev)
        self.final = self._buf()
        # regions=True (tests): go straight to the general region plan; False: never use it
        if regions:
            if not self._build_regions(region_plan(self.g, max_nodes or GRAPH_MAX_NODES), device):
                raise DspConfigError("this graph has no region plan")
            return
        # max_nodes (tests): cut the graph as if a kernel held only that many nodes
        plan = fused_plan(self.g) if fused is not False and max_nodes is None else None
        if plan is not None:
            eng = Engine(channels, max_frames, device=device, tile_channels=tile_channels)
            try:
                eng.set_graph(*plan)
                self.fused = eng
            except DspfxError as e:
                eng.close()
                if fused or e.status != ERR_UNSUPPORTED:
                    raise
        elif fused:
            raise DspConfigError("this graph cannot be fused into one kernel")
        if self.fused is not None:
            self.util = self.fused
            return
        # FIR / Fuzz nodes in series with fusable sub-graphs: one generated kernel per segment
        self.series = []                   # [(engine, out buffer)]
        steps = (segment_plan(self.g, max_nodes) if max_nodes is not None else series_plan(self.g)) if fused is None else None
        if steps is not None:
            try:
                for kind, *what in steps:
                    # "node": the hop into it is the previous segment's Output average; "node_hop": a raw handover
                    eng = Engine(channels, max_frames, link_flags=LINK_INPUT if kind == "node_hop" else 0, device=device,
                                 tile_channels=tile_channels)
                    self.series.append((eng, self._buf()))
                    if kind == "graph":
                        specs, links, in_ref, in2_ref = what
                        self.series_kind.append((kind, in_ref, in2_ref if any(l[0] == GRAPH_INPUT2 for l in links) else None))
                        eng.set_graph(specs, links)
                    else:
                        self.series_kind.append((kind, what[1], None))
                        eng.set_chain([what[0]])
            except DspfxError as e:
                for eng, _ in self.series:
                    eng.close()
                self.series, self.series_kind = [], []
                if e.status != ERR_UNSUPPORTED:
                    raise
        if self.series:
            self.util = self.series[0][0]
            return
        # anything else: regions with several input / output blocks each (region_plan), FIR / Fuzz nodes in between
        if fused is None and regions is not False and self._build_regions(region_plan(self.g, max_nodes or GRAPH_MAX_NODES), device):
            return
        self.runs, self.run_of = plan_runs(self.g)
        for r in self.runs:
            head = r.nodes[0]
            side_node = next((m for m in r.nodes if m.side), None)
            flags = run_link_flags(r)
            r.engine = Engine(channels, max_frames, link_flags=flags, device=device, tile_channels=tile_channels)
            r.engine.set_chain([m.spec for m in r.nodes])
            r.out = self._buf()
            if len(head.main) > 1:
                r.scratch_main = self._buf()
            if side_node is not None and len(side_node.side) > 1:
                r.scratch_side = self._buf()
        self.util = self.runs[0].engine if self.runs else Engine(channels, max_frames, device=device,
                                                                  tile_channels=tile_channels)

    def _buf(self):
        return self.torch.empty(self.B * self.N, dtype=self.torch.float32, device=self.dev)

    def _build_regions(self, steps, device) -> bool:
        """Engines and buffers for a region plan; False (nothing kept) when there is no plan or a kernel cannot be had."""
        if not steps:
            return False
        built = []
        try:
            for kind, *what in steps:
                if kind == "region":
                    specs, links, in_refs, n_out = what
                    eng = Engine(self.N, self.B, link_flags=0, device=device, tile_channels=self.tile)
                    built.append(dict(kind=kind, eng=eng, ins=in_refs, outs=[self._buf() for _ in range(n_out)]))
                    eng.set_graph(specs, links)
                elif kind == "node":
                    spec, main_refs, ctl_refs = what
                    flags = LINK_INTERNAL | (LINK_INPUT if len(main_refs) == 1 else 0)     # k links: averaged here, taken raw
                    eng = Engine(self.N, self.B, link_flags=flags, device=device, tile_channels=self.tile)
                    built.append(dict(kind=kind, eng=eng, main=main_refs, ctl=ctl_refs, outs=[self._buf()],
                                      scratch=self._buf() if len(main_refs) > 1 else None))
                    eng.set_chain([spec])
                else:
                    built.append(dict(kind=kind, eng=None, refs=what[0], outs=[self.final]))
        except DspfxError as e:
            for st in built:
                if st["eng"] is not None:
                    st["eng"].close()
            if e.status != ERR_UNSUPPORTED:
                raise
            return False
        self.regions = built
        self.util = next(st["eng"] for st in built if st["eng"] is not None) if any(st["eng"] for st in built) else \
            Engine(self.N, self.B, device=device, tile_channels=self.tile)
        return True

    def describe(self) -> str:
        if self.fused is not None:
            return "one kernel: " + " | ".join(l for l in self.fused.describe().splitlines() if l.startswith("stage"))
        if self.regions:
            out = []
            for k, st in enumerate(self.regions):
                if st["eng"] is None:
                    out.append(f"step {k}: Output node average of {len(st['refs'])} blocks")
                    continue
                what = f"region, {len(st['ins'])} in / {len(st['outs'])} out" if st["kind"] == "region" else "node"
                out.append(f"step {k} ({what}): " + " | ".join(l for l in st["eng"].describe().splitlines() if l.startswith("stage")))
            return "\n".join(out)
        if self.series:
            return "\n".join(f"segment {k}: " + " | ".join(l for l in eng.describe().splitlines() if l.startswith("stage"))
                             for k, (eng, _) in enumerate(self.series))
        lines = []
        for k, r in enumerate(self.runs):
            stage = [l for l in r.engine.describe().splitlines() if l.startswith("stage")]
            lines.append(f"run {k}: nodes {[m.id for m in r.nodes]}: " + " | ".join(stage))
        return "\n".join(lines)

    def _source(self, nid: int, x):
        if nid == ZERO:
            return self.zeros
        n = self.g.nodes[nid]
        if n.typename == "input":
            return x
        return self.run_of[nid].out

    def process(self, x, n_frames: Optional[int] = None, stream: int = 0):
        nf = self.B if n_frames is None else int(n_frames)
        if self.fused is not None:
            self.fused.process(self.zeros if x is None else x, out=self.final, n_frames=nf, stream=stream)
            return self.final
        if self.regions:
            x0 = self.zeros if x is None else x

            def at(ref):
                return x0 if ref == -1 else self.zeros if ref is None else self.regions[ref[0]]["outs"][ref[1]]

            for st in self.regions:
                if st["kind"] == "region":
                    st["eng"].process_io([at(r) for r in st["ins"]], st["outs"], nf, stream=stream)
                elif st["kind"] == "node":
                    refs = st["main"]
                    if len(refs) == 1:
                        src = at(refs[0])
                    elif not refs:
                        src = self.zeros
                    else:
                        st["eng"].link_average([at(r) for r in refs], st["scratch"], nf, stream)
                        src = st["scratch"]
                    ctl = {(0, k): at(r) for k, r in st["ctl"].items()}
                    st["eng"].process(src, out=st["outs"][0], n_frames=nf, stream=stream, ctl=ctl or None)
                else:
                    self.util.link_average([at(r) for r in st["refs"]], self.final, nf, stream)
            return self.regions[-1]["outs"][0]
        if self.series:
            x0 = self.zeros if x is None else x

            def block(ref):
                return x0 if ref == -1 else self.series[ref][1]

            for (eng, out), (kind, in_ref, in2_ref) in zip(self.series, self.series_kind):
                eng.process(block(in_ref), out=out, side=None if in2_ref is None else block(in2_ref), n_frames=nf, stream=stream)
            return self.series[-1][1]
        for nid in self.g.order:
            n = self.g.nodes[nid]
            if n.spec is None:
                continue
            r = self.run_of[nid]
            # a run is launched when its LAST node comes up: by then every producer of every node in it has run
            # (a buffer that feeds a side / slider port, or several consumers, always ends its own run)
            if r.nodes[-1].id != nid:
                continue
            head = r.nodes[0]
            if len(head.main) == 1:
                src = self._source(head.main[0], x)
            elif not head.main:
                src = self.zeros          # an unconnected port stays zeroed (node.rs:288); generators ignore it
            else:
                r.engine.link_average([self._source(s, x) for s in head.main], r.scratch_main, nf, stream)
                src = r.scratch_main
            side = None
            side_node = next((m for m in r.nodes if m.side), None)
            if side_node is not None:
                if len(side_node.side) == 1:
                    side = self._source(side_node.side[0], x)
                else:
                    r.engine.link_average([self._source(s, x) for s in side_node.side], r.scratch_side, nf, stream)
                    side = r.scratch_side
            ctl = {}
            for k, m in enumerate(r.nodes):
                for slider, srcs in m.ctl.items():
                    if len(srcs) != 1:   # node.rs:162-194 would average them; dspfx_process_ctl takes one signal per port
                        raise DspConfigError(f"slider port {slider} of node {m.id} averages several links")
                    ctl[(k, slider)] = self._source(srcs[0], x)
            r.engine.process(src, out=r.out, side=side, n_frames=nf, stream=stream, ctl=ctl or None)
        out_node = self.g.nodes[self.g.outputs[0]]
        self.util.link_average([self._source(s, x) for s in out_node.main], self.final, nf, stream)
        return self.final

    def tune_placement(self, x, n_frames: Optional[int] = None, stream: int = 0):
        """Re-tune the delay rings' placement of every generated kernel against the blocks it will really read and write
        (`dspfx_tune_placement`; resets DSP state).  `x` is the Input block the host will keep passing to `process`."""
        nf = min(self.B, 128) if n_frames is None else int(n_frames)
        if self.regions:
            return                      # region kernels exchange several blocks: not tuned (dspfx_tune_placement skips them too)
        if self.fused is not None:
            self.fused.tune_placement(x, self.final, nf, stream=stream)
            return
        x0 = self.zeros if x is None else x
        for (eng, out), (kind, in_ref, in2_ref) in zip(self.series, self.series_kind):
            src = x0 if in_ref == -1 else self.series[in_ref][1]
            side = None if in2_ref is None else (x0 if in2_ref == -1 else self.series[in2_ref][1])
            eng.tune_placement(src, out, nf, side=side, stream=stream)

    def close(self):
        if self.fused is not None:
            self.fused.close()
        for st in self.regions:
            if st["eng"] is not None:
                st["eng"].close()
        for eng, _ in self.series:
            eng.close()
        for r in self.runs:
            r.engine.close()
