#!/usr/bin/env python3
"""Export the repository's golden vectors as `cases.json` for the reference-side harness `golden_dump.rs` (oracle/pin_kit/README.md).

Every case is a DSPConfig document (the reference's own File > Save format, runtime.rs:44-48) + the Input node's samples as f32
bit patterns + the expected Output signal (`want`, bit patterns) + the bar the parity tests apply to it:

  * tests/golden/graphs/*.npz        whole saved graphs (fan-in, wet/dry, control ports, FIR cabinets): doc / x / y as stored;
  * tests/golden/*.npz               1-channel chains and single nodes: written as input -> chain -> output documents by
                                     dsp_stuff_amd.config.dump_dspconfig (the writer the GUI-format tests check), expected output =
                                     the oracle's node-by-node evaluation of THAT document (oracle/graph_eval.py: every hop of a real
                                     graph divides by f32(1.0001), the Output node's included) -- for the goldens stored with link
                                     scaling on (link_flags 3) this is cross-checked here against the stored vector + the Output hop;
  * random DAGs                      24 seeded graphs of tests/graphs.py (every node kind, control ports fed by other nodes, fan-in);
  * menu-fresh nodes                 a Reverb built by `NodeStatic::new` (make_buffer's ring under a 0.5 s slider, reverb.rs:44-52) and a
                                     BiQuad on its initial filter (biquad.rs:48-60): `fresh` tells the harness to use the NODES table.

The expected values come from the CPU oracle (PARITY UNPINNED: the reference ships no vectors) -- comparing the reference's own
output against them is exactly what pins it.  Also checks that the biquad probe hard-coded in golden_dump.rs separates the candidate
operation orders of DirectForm1::run.   usage: python oracle/pin_kit/export_cases.py [out.json]"""
import glob
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
for p in (ROOT, os.path.join(ROOT, "tests"), os.path.join(ROOT, "oracle")):
    sys.path.insert(0, p)
from __graft_entry__ import load_package  # noqa: E402

E = load_package()
from dsp_stuff_amd import config as CFG, graph as G  # noqa: E402
import graph_eval  # noqa: E402
import oracle as O  # noqa: E402
from golden_util import bar_for, load_all  # noqa: E402

F = np.float32
BIQUAD_PROBE_COEFFS = [-1.7990895, 0.81783146, 0.0046772375, 0.009354475, 0.0046772375]          # a1 a2 b0 b1 b2 (golden_dump.rs: probes())
BIQUAD_PROBE_X = [0.18588203191757202, -0.4798051118850708, 0.6797630190849304, -0.9063479900360107, 0.3373520076274872,
                  0.7703438401222229, -0.12345679104328156, 0.5555555820465088]


def bits(a):
    return np.ascontiguousarray(a, F).view(np.uint32).tolist()


def spec_from_desc(d):
    """oracle descriptor (tests/golden meta) -> the package's NodeSpec"""
    k, p = d["kind"], list(d.get("params") or [])
    if k == O.REVERB:
        # a golden's explicit ring length: the document carries it as the seconds slider (restore -> refresh_seconds, reverb.rs:55-71)
        D = int(d["delay_len"])
        s = float(F(D / 48000.0))
        while int(F(s) * F(48000.0)) < D:
            s = float(np.nextafter(F(s), F(2)))
        assert max(int(F(s) * F(48000.0)), 128) == D, (D, s)
        return E.Reverb(seconds=s, decay=p[0])
    if k == O.FIR:
        return E.NodeSpec(E.FIR, [], mode=d.get("mode", 0), taps_reversed=np.asarray(d["taps_reversed"], np.float64))
    return E.NodeSpec(k, p, mode=d.get("mode", 0))


def biquad_orders(c, xs):
    a1, a2, b0, b1, b2 = [F(v) for v in c]
    out = {}
    for order in ("left_to_right", "b_terms_minus_a_terms", "right_to_left", "fused_left_to_right"):
        x1 = x2 = y1 = y2 = F(0)
        ys = []
        for x in xs:
            x = F(x)
            if order == "left_to_right":            # b0*x + b1*x1 + b2*x2 - a1*y1 - a2*y2, as written (the oracle's choice: SURVEY 8c)
                y = F(F(F(F(F(b0 * x) + F(b1 * x1)) + F(b2 * x2)) - F(a1 * y1)) - F(a2 * y2))
            elif order == "b_terms_minus_a_terms":
                y = F(F(F(F(b0 * x) + F(b1 * x1)) + F(b2 * x2)) - F(F(a1 * y1) + F(a2 * y2)))
            elif order == "right_to_left":
                y = F(F(b0 * x) + F(F(b1 * x1) + F(F(b2 * x2) - F(F(a1 * y1) + F(a2 * y2)))))
            else:                                    # every product fused into the running sum (one rounding each)
                acc = F(np.float64(b0) * np.float64(x))
                acc = F(np.float64(b1) * np.float64(x1) + np.float64(acc))
                acc = F(np.float64(b2) * np.float64(x2) + np.float64(acc))
                acc = F(np.float64(acc) - np.float64(a1) * np.float64(y1))
                y = F(np.float64(acc) - np.float64(a2) * np.float64(y2))
            x2, x1 = x1, x
            y2, y1 = y1, y
            ys.append(y)
        out[order] = bits(np.array(ys, F))
    return out


def build_cases(page_round=False):
    """page_round: evaluate every delay ring under the OTHER reading of rivulet (granted view = whole 4 KiB pages) -- what
    tools/compare_pin.py switches to when the harness' probe says so.
    every case with its expected output (`want`) and bar -- tools/compare_pin.py calls this; cases.json carries only what the
    Rust side needs (name, doc, fresh, x)"""
    cases = []
    # ---- whole graphs
    for path in sorted(glob.glob(os.path.join(ROOT, "tests", "golden", "graphs", "*.npz"))):
        z = np.load(path)
        doc = json.loads(str(z["doc"]))
        x, y = z["x"], z["y"]
        if page_round:
            y = graph_eval.run_graph(G.Graph(json.dumps(doc), True), x)
        cases.append(dict(name="graph_" + os.path.basename(path)[:-4], doc=doc, fresh={}, x=[bits(x[:, c]) for c in range(x.shape[1])],
                          want=[bits(y[:, c]) for c in range(y.shape[1])], bar=dict(kind="ulp", ulp=4), source="tests/golden/graphs/" + os.path.basename(path)))
    # ---- chains and single nodes
    for g in load_all():
        chain = [spec_from_desc(d) for d in g["descs"]]
        text = CFG.dump_dspconfig(chain, faithful_lowpass_bug=False)
        x = np.asarray(g["x"], F)[:, None]
        y = graph_eval.run_graph(G.Graph(text, page_round), x)
        if g["link_flags"] == 3 and not page_round:     # the stored vector is the chain with every hop but the Output node's: add that one and compare
            hop = (F(0.0) + g["y"]) / E.link_divisor(1)
            assert np.array_equal(hop.astype(F).view(np.uint32), y[:, 0].view(np.uint32)), g["name"]
        b = bar_for(g["name"])
        bar = dict(kind="ulp", ulp=int(b)) if b is not None else (dict(kind="rel_peak", tol=4e-6) if g["name"].startswith("distort") else dict(kind="rel_rms", tol=1e-6))
        cases.append(dict(name="chain_" + g["name"], doc=json.loads(text), fresh={}, x=[bits(x[:, 0])], want=[bits(y[:, 0])], bar=bar,
                          source="tests/golden/%s.npz as a document; expected = oracle/graph_eval.py on it" % g["name"]))
    # ---- random DAGs (tests/graphs.py random_dag: fan-in, fan-out, unplugged ports, sliders driven by other nodes, generators, envelope
    # followers): 16 of the exact-arithmetic kinds, 8 with the libm kinds (tanh / sin / atan modes, overdrive, chebyshev, sine generator)
    import graphs
    want_exact, want_libm, seed = 16, 8, 7000
    while want_exact or want_libm:
        libm = want_exact == 0
        text = graphs.random_dag(seed, 8 + seed % 5, libm=libm)
        seed += 1
        x = O.noise(0x5EED00CC + seed, [0], np.arange(384))
        y = graph_eval.run_graph(G.Graph(text, page_round), x)
        if not np.isfinite(y).all() or np.abs(y).max() == 0:
            continue
        cases.append(dict(name="dag_%s_%d" % ("libm" if libm else "exact", seed - 1), doc=json.loads(text), fresh={}, x=[bits(x[:, 0])], want=[bits(y[:, 0])],
                          # (libm DAGs: a 1-ulp difference between two math libraries at one node is amplified by whatever follows it -- clippers,
                          # squarers, generators whose frequency it drives; the per-node ulp bars are the chain_* cases', these check the wiring)
                          bar=dict(kind="rel_peak", tol=1e-4) if libm else dict(kind="ulp", ulp=4), source="tests/graphs.py random_dag(%d); expected = oracle/graph_eval.py" % (seed - 1)))
        if libm:
            want_libm -= 1
        else:
            want_exact -= 1
    # ---- menu-fresh nodes (NodeStatic::new): make_buffer's 128-sample ring, the initial DirectForm1
    for name, node, title in (("fresh_reverb", E.Reverb(), "Reverb"), ("fresh_biquad", E.BiQuad(), "Biquad")):
        text = CFG.dump_dspconfig([node], faithful_lowpass_bug=False)
        doc = json.loads(text)
        nid = [n["id"] for n in doc["nodes"] if n["typename"] not in ("input", "output")][0]
        x = O.noise(0x5EED00BB, [0], np.arange(1024))
        spec_graph = G.Graph(text, page_round)
        if name == "fresh_reverb":   # the parsed document would restore (24000 samples); the fresh node sits on 128 (1024 if rivulet page-rounds: the probe tells)
            for n in spec_graph.nodes.values():
                if n.spec is not None and n.spec.kind == E.REVERB:
                    n.spec = E.Reverb(page_round=page_round)
        y = graph_eval.run_graph(spec_graph, x)
        cases.append(dict(name=name, doc=doc, fresh={str(nid): title}, x=[bits(x[:, 0])], want=[bits(y[:, 0])], bar=dict(kind="ulp", ulp=1),
                          source="NodeStatic::new via the NODES table; expected = oracle on the menu-fresh descriptor"))
    return cases


def probe_expectations():
    probe = biquad_orders(BIQUAD_PROBE_COEFFS, BIQUAD_PROBE_X)
    assert len({tuple(v) for v in probe.values()}) == len(probe), "the biquad probe no longer separates the candidate orders"
    return dict(biquad=dict(coeffs=bits(np.array(BIQUAD_PROBE_COEFFS, F)), x=bits(np.array(BIQUAD_PROBE_X, F)), candidates=probe,
                                       oracle_uses="left_to_right"),
                           rivulet_view_len=dict(exact={str(n): n for n in (128, 1000, 1024, 4800, 24000, 24576, 48000)},
                                                 page_rounded={str(n): -(-n // 1024) * 1024 for n in (128, 1000, 1024, 4800, 24000, 24576, 48000)},
                                                 oracle_default="exact (dspfx_delay_len(seconds, page_round = 0)); page_round = 1 is the other reading"))


def main():
    out_path = sys.argv[1] if len(sys.argv) > 1 else os.path.join(HERE, "cases.json")
    cases = build_cases()
    probe_expectations()
    doc = dict(schema=1, made_by="oracle/pin_kit/export_cases.py",
               cases=[{k: c[k] for k in ("name", "doc", "fresh", "x")} for c in cases])
    with open(out_path, "w") as f:
        json.dump(doc, f, separators=(",", ":"))
    print("%d cases -> %s (%d KiB)" % (len(cases), out_path, os.path.getsize(out_path) >> 10))


if __name__ == "__main__":
    main()
