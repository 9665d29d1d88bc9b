"""The C++ saved-graph importer (include/dspfx_graph.hpp: JSON parser, port / link resolution, Mux / Demux routing,
topological order, the plan for dspfx_graph_set) against the Python mirror: same plan for every test graph (no GPU
needed), and on the GPU box the graphs run from C++ match the committed golden vectors."""
import json
import os
import subprocess

import numpy as np
import pytest

import graphs

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
EXE = os.path.join(ROOT, "tests", "cpp", "test_graph")


def _build():
    cs = os.path.join(ROOT, "dsp-stuff_amd", "csrc")
    subprocess.check_call(["g++", "-std=c++17", "-O1", "-Wall", "-Wextra", "-o", EXE, os.path.join(ROOT, "tests", "cpp", "test_graph.cpp"),
                           f"-L{cs}", "-ldspfx", f"-Wl,-rpath,{cs}", "-L/opt/rocm/lib", "-Wl,-rpath,/opt/rocm/lib",
                           "-Wl,--allow-shlib-undefined"])


def _cpp_plan(tmp_path, text):
    p = tmp_path / "doc.json"
    p.write_text(text)
    r = subprocess.run([EXE, str(p), "--plan"], capture_output=True, text=True)
    assert r.returncode == 0, r.stdout + r.stderr
    return r.stdout.strip().splitlines()


def test_cpp_importer_plans_what_the_python_mirror_plans(dspfx, tmp_path):
    from dsp_stuff_amd import graph as G
    _build()
    docs = [graphs.diamond(), graphs.lfo_tremolo(), graphs.fan_in_three(), graphs.routing("A", "B"), graphs.routing("B", "A"),
            graphs.routing("A", "A")] + [graphs.random_dag(s, 8 + s % 9, libm=bool(s % 2)) for s in range(24)]
    for text in docs:
        specs, links = G.fused_plan(G.Graph(text))
        want = []
        for sp in specs:
            p = [float(np.float32(v)) for v in sp.params] + [None] * 6
            want.append((sp.kind, sp.mode, p[:len(sp.params)], int(sp.delay_len)))
        got_nodes, got_links = [], []
        for line in _cpp_plan(tmp_path, text):
            w = line.split()
            if w[0] == "node":
                got_nodes.append((int(w[1]), int(w[2]), [float.fromhex(v) for v in w[3:9]], int(w[9])))
            else:
                got_links.append(tuple(int(v) for v in w[1:4]))
        assert got_links == [tuple(l) for l in links]
        assert len(got_nodes) == len(want)
        for (k, m, p, d), (wk, wm, wp, wd) in zip(got_nodes, want):
            assert (k, m, d) == (wk, wm, wd) and p[:len(wp)] == wp, (k, m, p, d, wk, wm, wp, wd)
    # graphs that need cutting: the same series of engines as graph.py's segment_plan
    E = dspfx
    series = [graphs.cab_rig(), graphs.cab_rig(dry=True), graphs.cab_rig(cut="fuzz")] + \
             [graphs.long_rig(s, 12, fir_at=5 if s % 2 else None, dry_mix=s >= 5) for s in range(10)]
    for text in series:
        steps = G.series_plan(G.Graph(text))
        want = []
        for kind, *what in steps:
            in_ref, in2_ref = (what[2], what[3]) if kind == "graph" else (what[1], None)
            if kind == "graph" and not any(l[0] == E.GRAPH_INPUT2 for l in what[1]):
                in2_ref = None
            want.append("step %s %d %d" % (kind, in_ref, -2 if in2_ref is None else in2_ref))
            specs = what[0] if kind == "graph" else [what[0]]
            for sp in specs:
                p0 = float(np.float32(sp.params[0])) if sp.params else None
                want.append(("node", sp.kind, sp.mode, p0, int(sp.delay_len), 0 if sp.taps_reversed is None else len(sp.taps_reversed)))
            if kind == "graph":
                want += [("link",) + tuple(l) for l in what[1]]
        got = []
        for line in _cpp_plan(tmp_path, text):
            w = line.split()
            if w[0] == "step":
                got.append(line)
            elif w[0] == "node":
                got.append(("node", int(w[1]), int(w[2]), float.fromhex(w[3]), int(w[4]), int(w[5])))
            else:
                got.append(("link",) + tuple(int(v) for v in w[1:4]))
        assert len(got) == len(want)
        for a, b in zip(got, want):
            if isinstance(b, tuple) and b[0] == "node":
                assert a[:3] == b[:3] and a[4:] == b[4:] and (b[3] is None or a[3] == b[3]), (a, b)
            else:
                assert a == b, (a, b)
    assert _cpp_plan(tmp_path, graphs.cab_rig(bypass=True)) == ["run by run"]         # neither mirror can cut this one
    bad = tmp_path / "bad.json"
    bad.write_text('{"nodes": [], "links": []')
    r = subprocess.run([EXE, str(bad), "--plan"], capture_output=True, text=True)
    assert r.returncode == 3 and "not JSON" in r.stdout


def test_cpp_region_plan_equals_the_python_one(dspfx, tmp_path):
    """SavedGraph::region_plan (any graph as a few generated kernels exchanging several blocks) against graph.py's."""
    from dsp_stuff_amd import graph as G
    E = dspfx
    _build()
    docs = [graphs.random_dag(s, n) for s, n in ((1, 40), (2, 40), (3, 30), (11, 23), (13, 5))] + \
           [graphs.around_fir_and_fuzz(), graphs.cab_rig(bypass=True), graphs.cab_rig(cut="fuzz"), graphs.diamond(), graphs.routing("B", "A")]
    for text in docs:
        p = tmp_path / "doc.json"
        p.write_text(text)
        r = subprocess.run([EXE, str(p), "--plan-regions"], capture_output=True, text=True)
        assert r.returncode == 0, r.stdout + r.stderr
        got = r.stdout.strip().splitlines()
        want = []

        def rr(ref):
            return (-1, 0) if ref == -1 else (-2, 0) if ref is None else tuple(ref)

        for kind, *what in G.region_plan(G.Graph(text)):
            if kind == "region":
                specs, links, in_refs, n_out = what
                want.append("step region %d" % n_out)
                want += ["in %d %d" % rr(x) for x in in_refs]
            elif kind == "node":
                spec, main_refs, ctl_refs = what
                specs, links = [spec], []
                want.append("step node 0")
                want += ["main %d %d" % rr(x) for x in main_refs]
                want += ["ctl %d %d %d" % ((k,) + rr(x)) for k, x in sorted(ctl_refs.items())]
            else:
                specs, links = [], []
                want.append("step output 0")
                want += ["main %d %d" % rr(x) for x in what[0]]
            for sp in specs:
                want.append(("node", sp.kind, sp.mode, int(sp.delay_len), 0 if sp.taps_reversed is None else len(sp.taps_reversed)))
            want += ["link %d %d %d" % tuple(l) for l in links]
        assert len(got) == len(want), (got, want)
        for a, b in zip(got, want):
            if isinstance(b, tuple):
                w = a.split()
                assert w[0] == "node" and (int(w[1]), int(w[2]), int(w[4]), int(w[5])) == b[1:], (a, b)
            else:
                assert a == b, (a, b)


@pytest.mark.gpu
def test_cpp_region_plan_runs_on_the_gpu(dspfx, tmp_path):
    """The region plan driven from C++ through dspfx_process_io (blocks in page-locked host memory): the golden graphs and
    a 40-node random DAG against the oracle."""
    import graph_eval
    import oracle as O
    from dsp_stuff_amd import graph as G
    from test_graph_cpu import _graph_goldens
    _build()
    cases = [(name, doc, x, y) for name, doc, x, y in _graph_goldens()]
    text = graphs.random_dag(2, 40)
    x = O.noise(0x5EED0042, np.arange(3), np.arange(384))
    cases.append(("random40", text, x, graph_eval.run_graph(G.Graph(text), x)))
    for name, doc, x, y in cases:
        (tmp_path / "doc.json").write_text(doc)
        x.astype(np.float32).tofile(tmp_path / "x.f32")
        y.astype(np.float32).tofile(tmp_path / "y.f32")
        loose = ["--fir-tolerance"] if ("fir" in name or "cab" in name) else []
        r = subprocess.run([EXE, str(tmp_path / "doc.json"), str(tmp_path / "x.f32"), str(tmp_path / "y.f32"), str(x.shape[1]),
                            str(x.shape[0]), "--regions"] + loose, capture_output=True, text=True)
        assert r.returncode == 0 and "region plan" in r.stdout, (name, r.stdout, r.stderr)


@pytest.mark.gpu
def test_cpp_importer_runs_the_golden_graphs(dspfx, tmp_path):
    from test_graph_cpu import _graph_goldens
    _build()
    ran = 0
    for name, doc, x, y in _graph_goldens():
        (tmp_path / "doc.json").write_text(doc)
        x.astype(np.float32).tofile(tmp_path / "x.f32")
        y.astype(np.float32).tofile(tmp_path / "y.f32")
        loose = ["--fir-tolerance"] if ("fir" in name or "cab" in name) else []
        r = subprocess.run([EXE, str(tmp_path / "doc.json"), str(tmp_path / "x.f32"), str(tmp_path / "y.f32"), str(x.shape[1]),
                            str(x.shape[0])] + loose, capture_output=True, text=True)
        assert r.returncode == 0, (name, r.stdout, r.stderr)
        ran += 1
    assert ran >= 8


def test_cpp_exporter_writes_what_the_importers_read(dspfx, tmp_path):
    """dump_dspconfig in C++ (a chain written the way File > Save would): the Python importer reads back the chain, the C++
    importer plans it, and the LowPass cfg_name quirk (saved as "high_pass") is reproduced on request."""
    from dsp_stuff_amd import config
    E = dspfx
    _build()
    r = subprocess.run([EXE, "--dump-chain"], capture_output=True, text=True)
    assert r.returncode == 0
    chain, info = config.load_dspconfig(r.stdout)
    want = [E.BiQuad(1.0, -1.8, 0.81, 0.0025, 0.005, 0.0025), E.LowPass(0.3), E.Distort(3.0, E.TANH), E.Reverb(seconds=0.5, decay=0.4),
            E.Mix(0.25), E.Overdrive(2.0, 0.5, 0.75), E.Fir([0.5, 0.25, -0.125], E.FIR_AVERAGE), E.Envelope(4.0, 100.0), E.Gain(0.1)]
    assert len(chain) == len(want) and info["side_from_input"]
    for a, b in zip(chain, want):
        assert (a.kind, a.mode, a.delay_len) == (b.kind, b.mode, b.delay_len)
        assert [np.float32(v) for v in a.params] == [np.float32(v) for v in b.params]
        if b.taps_reversed is not None:
            assert np.array_equal(np.asarray(a.taps_reversed, np.float64), np.asarray(b.taps_reversed, np.float64))
    faithful = subprocess.run([EXE, "--dump-chain-faithful"], capture_output=True, text=True).stdout
    chain2, _ = config.load_dspconfig(faithful)
    assert chain2[1].kind == E.HIGH_PASS                                  # low_pass.rs:9: a saved LowPass restores as a HighPass
    assert json.loads(r.stdout)["nodes"][2]["typename"] == "low_pass" and json.loads(faithful)["nodes"][2]["typename"] == "high_pass"
