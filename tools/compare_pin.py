#!/usr/bin/env python3
"""Hold the reference's own outputs (pin_out.json, written by oracle/pin_kit/golden_dump.rs inside the reference tree) against the
repository's golden vectors, with the bars of the parity tests (tests/golden_util.py, tests/test_graph_gpu.py) -- the one step
that turns "parity unpinned" into "pinned" (oracle/pin_kit/README.md).

    python tools/compare_pin.py pin_out.json          per case: worst ulp / relative error and PASS / FAIL; the two probes read out;
                                                      exit status 0 only when every case is inside its bar
    python tools/compare_pin.py --emulate out.json    self-test of the kit without a Rust toolchain: writes the file golden_dump.rs
                                                      would write, produced by walking cases.json the way golden_dump.rs does (pipes
                                                      per link, ports by name, topological order, the Output node's hop) over the
                                                      CPU oracle's nodes -- then compare it like any other

Bars: `ulp` -- same finite pattern, every finite sample within N units of f32 ordering; `rel_peak` -- |diff| <= tol x max|want|
(Fuzz: block-global maxima); `rel_rms` -- RMS(diff) <= tol x RMS(want) (FIR against the f64 accumulation)."""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
KIT = os.path.join(ROOT, "oracle", "pin_kit")
for p in (ROOT, KIT, os.path.join(ROOT, "tests"), os.path.join(ROOT, "oracle")):
    sys.path.insert(0, p)
F = np.float32


def f32(bits):
    return np.array(bits, np.uint32).view(F)


def ulp(a, b):
    ia, ib = a.view(np.int32).astype(np.int64), b.view(np.int32).astype(np.int64)
    ia = np.where(ia < 0, -(ia & 0x7FFFFFFF), ia)
    ib = np.where(ib < 0, -(ib & 0x7FFFFFFF), ib)
    return np.abs(ia - ib)


def judge(got, want, bar):
    """-> (ok, text)"""
    if got.shape != want.shape:
        return False, "length %d, expected %d" % (got.size, want.size)
    fin = np.isfinite(want)
    if not np.array_equal(np.isfinite(got), fin):
        return False, "the finite / non-finite pattern differs"
    g, w = got[fin], want[fin]
    if g.size == 0:
        return True, "no finite samples (both sides agree)"
    if bar["kind"] == "ulp":
        d = int(ulp(g, w).max())
        return d <= bar["ulp"], "worst %d ulp (bar %d)" % (d, bar["ulp"])
    e = g.astype(np.float64) - w.astype(np.float64)
    if bar["kind"] == "rel_peak":
        r = float(np.abs(e).max() / max(np.abs(w).max(), 1e-30))
        return r <= bar["tol"], "max |diff| / peak %.3g (bar %.3g)" % (r, bar["tol"])
    r = float(np.sqrt(np.mean(e ** 2)) / max(np.sqrt(np.mean(w.astype(np.float64) ** 2)), 1e-30))
    return r <= bar["tol"], "relative RMS %.3g (bar %.3g)" % (r, bar["tol"])


def emulate(out_path):
    """golden_dump.rs, step by step, over the oracle's nodes"""
    from __graft_entry__ import load_package
    E = load_package()
    from dsp_stuff_amd import config as CFG
    import graph_eval
    import oracle as O
    from dsp_stuff_amd import graph as G
    import export_cases as X
    cases = json.load(open(os.path.join(KIT, "cases.json")))["cases"]
    L = O.lib()
    fresh_specs = {"Reverb": E.Reverb, "Biquad": E.BiQuad, "Gain": E.Gain}
    results = []
    for case in cases:
        doc = case["doc"]
        by_id = {n["id"]: n for n in doc["nodes"]}
        walkable = all(n["typename"] in CFG._TABLE or n["typename"] in ("input", "output") for n in doc["nodes"])
        ys = []
        for ch in case["x"]:
            x = f32(ch)
            if not walkable:          # mux / demux documents: the structural evaluation of oracle/graph_eval.py
                ys.append(graph_eval.run_graph(G.Graph(json.dumps(doc)), x[:, None])[:, 0])
                continue
            nodes = {}
            for n in doc["nodes"]:
                if n["typename"] in ("input", "output"):
                    continue
                title = case["fresh"].get(str(n["id"]))
                spec = fresh_specs[title]() if title else CFG._node_from_cfg(n["typename"], n["cfg"], False)
                nodes[n["id"]] = O.node_from_desc(spec.oracle_desc())
            order, done = [], set()
            while len(done) < len(doc["nodes"]):
                for n in doc["nodes"]:
                    if n["id"] not in done and all(l["lhs"][0] in done for l in doc["links"] if l["rhs"][0] == n["id"]):
                        done.add(n["id"])
                        order.append(n["id"])
            pname = lambda nid, which, pid: next(k for k, v in by_id[nid]["cfg"][which].items() if v == pid)
            y = np.empty_like(x)
            for f0 in range(0, x.size, 128):
                pipe = {}
                for nid in order:
                    n = by_id[nid]
                    ins = {}
                    for k, l in enumerate(doc["links"]):
                        if l["rhs"][0] == nid:
                            ins.setdefault(pname(nid, "inputs", l["rhs"][1]), []).append(pipe[k])
                    if n["typename"] == "input":
                        val = x[f0:f0 + 128]
                    elif n["typename"] == "output":
                        y[f0:f0 + 128] = graph_eval._avg(L, ins.get("in", []), 128)
                        continue
                    else:
                        _, _, main, ctls = CFG._TABLE[n["typename"]]
                        a = graph_eval._avg(L, ins.get(main, []), 128) if main else np.zeros(128, F)
                        b = graph_eval._avg(L, ins["b"], 128) if "b" in ins else (np.zeros(128, F) if n["typename"] in ("add", "mix") else None)
                        ctl = None
                        if any(c in ins for c in ctls):
                            ctl = [graph_eval._avg(L, ins[c], 128) if c in ins else None for c in ctls] + [None] * (3 - len(ctls))
                        val = nodes[nid].process(a, b, ctl)
                    for k, l in enumerate(doc["links"]):
                        if l["lhs"][0] == nid:
                            pipe[k] = val
            ys.append(y)
        results.append(dict(name=case["name"], y=[np.ascontiguousarray(v, F).view(np.uint32).tolist() for v in ys]))
    pe = X.probe_expectations()
    probes = dict(rivulet_view_len=[[int(k), v] for k, v in pe["rivulet_view_len"]["exact"].items()],
                  biquad_probe=dict(coeffs=pe["biquad"]["coeffs"], x=pe["biquad"]["x"], y=pe["biquad"]["candidates"][pe["biquad"]["oracle_uses"]]))
    json.dump(dict(schema=1, buf_size=128, results=results, probes=probes, emulated=True), open(out_path, "w"), separators=(",", ":"))
    print("emulated %d cases -> %s" % (len(results), out_path))


def compare(path):
    import export_cases as X
    out = json.load(open(path))
    assert out.get("schema") == 1 and out.get("buf_size") == 128, "not a golden_dump.rs result file (schema / BUF_SIZE)"
    got = {r["name"]: r["y"] for r in out["results"]}
    bad = 0
    pe = X.probe_expectations()
    lens = {str(n): v for n, v in out["probes"]["rivulet_view_len"]}
    reading = [k for k in ("exact", "page_rounded") if all(pe["rivulet_view_len"][k].get(n) == v for n, v in lens.items())]
    # the delay rings' length is the one thing the oracle takes as a parameter (D explicit): the expected vectors are
    # evaluated under the reading the reference's own rivulet reports
    cases = X.build_cases(page_round=(reading == ["page_rounded"]))
    for c in cases:
        if c["name"] not in got:
            print("MISSING  %s" % c["name"])
            bad += 1
            continue
        worst_ok, notes = True, []
        for ch, (g, w) in enumerate(zip(got[c["name"]], c["want"])):
            ok, text = judge(f32(g), f32(w), c["bar"])
            worst_ok = worst_ok and ok
            notes.append(text)
        if len(got[c["name"]]) != len(c["want"]):
            worst_ok, notes = False, ["%d channels, expected %d" % (len(got[c["name"]]), len(c["want"]))]
        bad += 0 if worst_ok else 1
        print("%s  %-32s %s" % ("PASS   " if worst_ok else "FAIL   ", c["name"], "; ".join(sorted(set(notes)))))
    print("probe    rivulet granted view length %s -> %s" % (lens, ("the `%s` reading: dspfx_delay_len(seconds, page_round = %d)" % (reading[0], reading[0] == "page_rounded"))
                                                                if reading else "NEITHER reading the oracle knows: restate reverb.rs:60-68's length from these numbers"))
    by = out["probes"]["biquad_probe"]["y"]
    match = [k for k, v in pe["biquad"]["candidates"].items() if v == by]
    print("probe    biquad::DirectForm1::run operation order -> %s" % (match[0] if match else "NONE of the candidate orders: restate biquad.rs:87 from the crate's source"))
    if reading == ["page_rounded"]:
        print("         (the engine's default is page_round = 0: hosts of THIS rivulet pass page_round = 1 -- dspfx_delay_len(seconds, 1), mode bit 0 of a REVERB descriptor)")
    if not reading or match != [pe["biquad"]["oracle_uses"]]:
        print("         (the oracle assumes `%s`: where the reference says otherwise, the oracle -- not the reference -- is what changes)" % pe["biquad"]["oracle_uses"])
        bad += 1
    print("%d case(s) outside their bar or probe(s) contradicting the oracle" % bad if bad else
          "every case inside its bar and both probes inside what the oracle implements: parity PINNED against the reference%s" % (" (EMULATED run: says nothing about the reference)" if out.get("emulated") else ""))
    return 1 if bad else 0


if __name__ == "__main__":
    if len(sys.argv) == 3 and sys.argv[1] == "--emulate":
        emulate(sys.argv[2])
        sys.exit(0)
    if len(sys.argv) != 2:
        print(__doc__)
        sys.exit(2)
    sys.exit(compare(sys.argv[1]))
