"""Import / export of the reference's saved graphs (`DSPConfig` JSON).

The reference GUI saves the graph as (dsp-stuff/src/runtime.rs:44-48, 560-564, 606-612):
    {"nodes": [{"id": N, "typename": cfg_name, "position": [x, y], "cfg": {...}}],
     "links": [{"lhs": [node_id, out_port_id], "rhs": [node_id, in_port_id]}]}
where a node's `cfg` holds its id, its port maps {"inputs": {name: port_id}, "outputs": {...}}
and every field tagged `save` (dsp-stuff-derive/src/lib.rs:233-293): slider atomics as numbers,
select enums as variant names, Fir's taps as the stored (time-reversed) f64 array.

This module turns such a file into the engine's chain descriptor when the graph is a linear
chain  input -> n0 -> ... -> output  (the shape of every BASELINE config), and writes chains
back out in the same format.  Graphs the fused engine cannot express raise DspConfigError:
fan-in/fan-out, control ports fed by links (`as_input` sliders), and node types outside the hot
path (mux, demux, muff, pitch, wave_view, spectrogram).

Reference quirks honoured on purpose:
  * LowPass declares cfg_name = "high_pass" (nodes/low_pass.rs:9), so a LowPass saved by the GUI
    comes back as a HighPass (RESTORE lookup, nodes/mod.rs:118-119); typename "low_pass" only
    appears in hand-written files and restores as LowPass.
  * restoring a reverb runs `refresh_seconds` (lib.rs:319-337), so its ring has
    max((seconds*48000) as usize, 128) samples -- not make_buffer()'s 128.
"""
from __future__ import annotations

import json
from typing import List, Tuple

import numpy as np

from . import (ADD, BIQUAD, CHEBYSHEV, DISTORT, DISTORT_MODES, ENVELOPE, FIR, FIR_AVERAGE, FIR_BALANCED, GAIN,
               HIGH_PASS, LOW_PASS, MIX, OVERDRIVE, REVERB, SIGNAL_GEN, NodeSpec, delay_len)


class DspConfigError(ValueError):
    pass


_FIR_MODES = {"Balanced": FIR_BALANCED, "Average": FIR_AVERAGE}
_UNSUPPORTED = {"mux", "demux", "muff", "pitch", "wave_view", "spectrogram"}
SIGNAL_MODES = ["Sine", "Triangle", "Square", "Constant"]   # signal_gen.rs:17-22
# typename -> (kind, saved slider fields in params order, main input port, as_input control ports)
_TABLE = {
    "gain": (GAIN, ["level"], "in", ["level"]),
    "biquad": (BIQUAD, ["a0", "a1", "a2", "b0", "b1", "b2"], "in", []),
    "low_pass": (LOW_PASS, ["ratio"], "in", []),
    "high_pass": (HIGH_PASS, ["ratio"], "in", []),
    "reverb": (REVERB, ["decay"], "in", []),
    "distort": (DISTORT, ["level"], "in", ["level"]),
    "overdrive": (OVERDRIVE, ["boost", "drive", "level"], "in", ["boost", "drive", "level"]),
    "chebyshev": (CHEBYSHEV, ["level_pos", "level_neg"], "in", []),
    "fir": (FIR, [], "in", []),
    "add": (ADD, [], "a", []),
    "mix": (MIX, ["ratio"], "a", ["ratio"]),
    "envelope": (ENVELOPE, ["attack", "release"], "in", []),
    "signal_gen": (SIGNAL_GEN, ["amplitude", "frequency"], None, ["amplitude", "frequency"]),   # a source: no main port
}


def _node_from_cfg(typename: str, cfg: dict, page_round: bool) -> NodeSpec:
    kind, fields, _, _ = _TABLE[typename]
    try:
        params = [float(cfg[f]) for f in fields]
    except KeyError as e:
        raise DspConfigError(f"{typename} node lacks saved field {e}") from None
    if kind == REVERB:
        # restore() sets the fields, then runs refresh_seconds (lib.rs:319-337): the ring has reverb.rs:58's length; the
        # seconds slider travels with the node (params[1]) so that a later slider store refreshes to the same length
        return NodeSpec(REVERB, params + [float(np.float32(cfg["seconds"]))], mode=int(bool(page_round)),
                        delay_len=delay_len(float(cfg["seconds"]), page_round))
    if kind == DISTORT:
        mode = cfg.get("mode", "SoftClip")
        if mode not in DISTORT_MODES:
            raise DspConfigError(f"unknown distort mode {mode!r}")
        return NodeSpec(DISTORT, params, mode=DISTORT_MODES.index(mode))
    if kind == SIGNAL_GEN:
        mode = cfg.get("mode", "Sine")
        if mode not in SIGNAL_MODES:
            raise DspConfigError(f"unknown signal_gen mode {mode!r}")
        return NodeSpec(SIGNAL_GEN, params, mode=SIGNAL_MODES.index(mode))
    if kind == FIR:
        mode = cfg.get("mode", "Balanced")
        if mode not in _FIR_MODES:
            raise DspConfigError(f"unknown fir mode {mode!r}")
        taps = np.asarray(cfg.get("taps", [1.0]), np.float64)     # stored reversed (fir.rs:163,168)
        if taps.ndim != 1 or taps.size == 0:
            raise DspConfigError("fir node needs a non-empty taps array")
        return NodeSpec(FIR, [], mode=_FIR_MODES[mode], taps_reversed=taps)
    return NodeSpec(kind, params)


def load_dspconfig(text: str, page_round: bool = False) -> Tuple[List[NodeSpec], dict]:
    """Parse a DSPConfig JSON string -> (chain, info).  info = {"order": [node ids], "side_from_input": bool}."""
    try:
        doc = json.loads(text)
        nodes = {int(n["id"]): n for n in doc["nodes"]}
        links = [(tuple(map(int, l["lhs"])), tuple(map(int, l["rhs"]))) for l in doc["links"]]
    except (KeyError, TypeError, ValueError) as e:
        raise DspConfigError(f"not a DSPConfig document: {e}") from None
    for n in nodes.values():
        if n["typename"] in _UNSUPPORTED:
            raise DspConfigError(f"node type {n['typename']!r} is outside the accelerated path")
        if n["typename"] not in _TABLE and n["typename"] not in ("input", "output"):
            raise DspConfigError(f"unknown node type {n['typename']!r}")
    inputs = [i for i, n in nodes.items() if n["typename"] == "input"]
    outputs = [i for i, n in nodes.items() if n["typename"] == "output"]
    gens = [i for i, n in nodes.items() if n["typename"] == "signal_gen"]
    # the chain's source is the input node, or -- in a patch without one -- a signal generator
    if not inputs and len(gens) == 1:
        inputs = gens
    elif gens:
        raise DspConfigError("a signal_gen node can only be the source of the chain (in place of the input node)")
    if len(inputs) != 1 or len(outputs) != 1:
        raise DspConfigError("expected exactly one input (or signal_gen) and one output node")

    def port_name(node_id, port_id, which):
        for name, pid in nodes[node_id]["cfg"].get(which, {}).items():
            if int(pid) == port_id:
                return name
        raise DspConfigError(f"link refers to unknown {which[:-1]} port {port_id} of node {node_id}")

    out_links, in_links = {}, {}
    for (ln, lp), (rn, rp) in links:
        if ln not in nodes or rn not in nodes:
            raise DspConfigError("link refers to a missing node")
        out_links.setdefault(ln, []).append((rn, port_name(rn, rp, "inputs")))
        in_links.setdefault((rn, port_name(rn, rp, "inputs")), []).append(ln)
    src = inputs[0]
    chain, order, side_from_input = [], [], False
    cur, seen = src, {src}
    if nodes[src]["typename"] == "signal_gen":
        for cp in _TABLE["signal_gen"][3]:
            if (src, cp) in in_links:
                raise DspConfigError(f"control port {cp!r} of node {src} is fed by a link (as_input modulation)")
        chain.append(_node_from_cfg("signal_gen", nodes[src]["cfg"], page_round))
        order.append(src)
    while True:
        nxt = out_links.get(cur, [])
        if cur == src:   # the input node may also feed port "b" of add/mix nodes (the engine's side input)
            main = [(n, p) for n, p in nxt if not (nodes[n]["typename"] in ("add", "mix") and p == "b")]
            side_from_input = len(main) != len(nxt)
            nxt = main
        if len(nxt) != 1:
            raise DspConfigError(f"node {cur} fans out to {len(nxt)} links: not a linear chain")
        nid, pname = nxt[0]
        if nid in seen:
            raise DspConfigError("graph has a cycle")
        seen.add(nid)
        n = nodes[nid]
        if n["typename"] == "output":
            if len(in_links.get((nid, pname), [])) != 1:
                raise DspConfigError("output node mixes several links: use the mix bus instead")
            break
        _, _, main_port, ctl_ports = _TABLE[n["typename"]]
        if pname != main_port:
            raise DspConfigError(f"chain enters node {nid} through port {pname!r}, expected {main_port!r}")
        if len(in_links[(nid, pname)]) != 1:
            raise DspConfigError(f"port {pname!r} of node {nid} averages several links (fan-in)")
        for cp in ctl_ports:
            if (nid, cp) in in_links:
                raise DspConfigError(f"control port {cp!r} of node {nid} is fed by a link (as_input modulation)")
        if (nid, "b") in in_links and in_links[(nid, "b")] != [src]:
            raise DspConfigError(f"port 'b' of node {nid} must be fed by the input node (side input)")
        chain.append(_node_from_cfg(n["typename"], n["cfg"], page_round))
        order.append(nid)
        cur = nid
    if len(seen) != len(nodes):
        raise DspConfigError("graph has nodes that are not on the input->output chain")
    return chain, {"order": order, "side_from_input": side_from_input}


_KIND_TO_TYPENAME = {GAIN: "gain", BIQUAD: "biquad", LOW_PASS: "high_pass",   # LowPass saves as "high_pass" (low_pass.rs:9)
                     HIGH_PASS: "high_pass", REVERB: "reverb", DISTORT: "distort", OVERDRIVE: "overdrive",
                     CHEBYSHEV: "chebyshev", FIR: "fir", ADD: "add", MIX: "mix", SIGNAL_GEN: "signal_gen", ENVELOPE: "envelope"}


def dump_dspconfig(chain: List[NodeSpec], seconds_for_delay=None, faithful_lowpass_bug: bool = True) -> str:
    """Write a chain in the reference's format (input -> chain -> output, fresh ids).
    `seconds_for_delay(delay_len) -> seconds` lets the caller choose the saved slider value of a
    reverb (default delay_len / 48000).  With faithful_lowpass_bug a LowPass is written the way
    the reference writes it (typename "high_pass"); pass False to write "low_pass"."""
    next_id = [0]

    def nid():
        next_id[0] += 1
        return next_id[0] - 1

    nodes, links = [], []
    if any(n.kind == SIGNAL_GEN for n in chain[1:]):
        raise DspConfigError("a signal_gen node can only be the first node of a chain")
    if any(n.kind in (ADD, MIX) for n in chain) and chain and chain[0].kind == SIGNAL_GEN:
        raise DspConfigError("add/mix take their 'b' port from the input node, which a generator-sourced patch lacks")
    in_id = in_port = prev = None
    if not (chain and chain[0].kind == SIGNAL_GEN):
        in_id, in_port = nid(), nid()
        nodes.append({"id": in_id, "typename": "input", "position": [0.0, 0.0],
                      # InputConfig (nodes/input.rs:32-38): id, selected_host, selected_device, outputs -- restore()
                      # unwraps the deserialisation, so every field must be there; an unknown host name simply
                      # opens no device (input.rs:197-204).  "ALSA" is cpal's default host on Linux.
                      "cfg": {"id": in_id, "selected_host": "ALSA", "selected_device": None, "outputs": {"out": in_port}}})
        prev = (in_id, in_port)
    for k, n in enumerate(chain):
        tn = _KIND_TO_TYPENAME[n.kind]
        if n.kind == LOW_PASS and not faithful_lowpass_bug:
            tn = "low_pass"
        _, fields, main_port, ctl_ports = _TABLE["low_pass" if n.kind == LOW_PASS else tn]
        node_id = nid()
        ins = {main_port: nid()} if main_port else {}
        if n.kind in (ADD, MIX):
            ins["b"] = nid()
        for cp in ctl_ports:
            ins[cp] = nid()
        outs = {"out": nid()}
        cfg = {"id": node_id, "inputs": ins, "outputs": outs}
        for f, v in zip(fields, n.params):
            cfg[f] = float(np.float32(v))
        if n.kind == REVERB:
            if len(n.params) > 1 and n.params[1] > 0:          # the node carries its seconds slider: save that
                cfg["seconds"] = float(np.float32(n.params[1]))
            else:
                cfg["seconds"] = float(seconds_for_delay(n.delay_len) if seconds_for_delay else np.float32(n.delay_len / 48000.0))
        if n.kind == DISTORT:
            cfg["mode"] = DISTORT_MODES[n.mode]
        if n.kind == SIGNAL_GEN:
            cfg["mode"] = SIGNAL_MODES[n.mode]
        if n.kind == FIR:
            cfg["mode"] = "Average" if n.mode == FIR_AVERAGE else "Balanced"
            cfg["file_name"] = None
            cfg["taps"] = [float(t) for t in np.asarray(n.taps_reversed, np.float64)]
        nodes.append({"id": node_id, "typename": tn, "position": [120.0 * (k + 1), 0.0], "cfg": cfg})
        if main_port:
            links.append({"lhs": list(prev), "rhs": [node_id, ins[main_port]]})
        if n.kind in (ADD, MIX):
            links.append({"lhs": [in_id, in_port], "rhs": [node_id, ins["b"]]})
        prev = (node_id, outs["out"])
    out_id, out_port = nid(), nid()
    nodes.append({"id": out_id, "typename": "output", "position": [120.0 * (len(chain) + 1), 0.0],
                  # OutputConfig (nodes/output.rs:32-38): id, selected_host, selected_device, inputs
                  "cfg": {"id": out_id, "selected_host": "ALSA", "selected_device": None, "inputs": {"in": out_port}}})
    links.append({"lhs": list(prev), "rhs": [out_id, out_port]})
    return json.dumps({"nodes": nodes, "links": links})
