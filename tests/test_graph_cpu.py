"""Saved-graph (DAG) support without a GPU: parsing, the cut into fused runs, and the CPU evaluation the GPU
tests compare against (checked here against the linear-chain oracle on a graph that IS a chain)."""
import json

import numpy as np
import pytest

import graph_eval
import graphs
import oracle as O

F = np.float32


def test_partition_into_runs(dspfx):
    from dsp_stuff_amd import graph as G
    g = G.Graph(graphs.diamond())
    runs, run_of = G.plan_runs(g)
    assert sorted(sorted(m.id for m in r.nodes) for r in runs) == [[1], [2, 3], [4], [5, 6]]
    flags = {tuple(m.id for m in r.nodes): G.run_link_flags(r) for r in runs}
    assert flags[(1,)] == dspfx.LINK_INTERNAL | dspfx.LINK_INPUT
    assert flags[(5, 6)] == dspfx.LINK_INTERNAL                      # two links into distort: pre-averaged, taken raw
    g = G.Graph(graphs.fan_in_three())
    runs, _ = G.plan_runs(g)
    flags = {tuple(m.id for m in r.nodes): G.run_link_flags(r) for r in runs}
    # distort (three links on "in") > mix (two links on "b") fuse: both ports are pre-averaged and taken raw
    assert flags[(4, 5)] == dspfx.LINK_INTERNAL | dspfx.LINK_SIDE_RAW
    assert flags[(6,)] == dspfx.LINK_INTERNAL and flags[(8,)] == dspfx.LINK_INTERNAL    # generator / nothing plugged in
    assert g.order.index(9) == len(g.order) - 1


def test_graph_eval_equals_chain_oracle_on_a_chain(dspfx):
    from dsp_stuff_amd import config, graph as G
    chain = [dspfx.BiQuad(), dspfx.Distort(3.0, dspfx.SOFT_CLIP), dspfx.Reverb(delay_samples=256, decay=0.5), dspfx.Gain(0.5)]
    g = G.Graph(config.dump_dspconfig(chain))
    x = O.noise(3, np.arange(3), np.arange(512))
    got = graph_eval.run_graph(g, x)
    # whole-graph semantics = every hop scaled, including the one into the Output node
    ref = O.run_channels([n.oracle_desc() for n in chain], x, 3)
    ref = (ref / O.link_divisor(1)).astype(F)
    assert np.array_equal(got, ref)


def test_unplugged_and_fan_in_semantics():
    from dsp_stuff_amd import graph as G
    g = G.Graph(graphs.fan_in_three())
    x = O.noise(5, np.arange(2), np.arange(256))
    y = graph_eval.run_graph(g, x)
    assert y.shape == x.shape and np.isfinite(y).all()
    # node 8 (biquad with nothing plugged in) contributes zeros, but still counts as a connected pipe of the output
    doc = json.loads(graphs.fan_in_three())
    doc["links"] = [l for l in doc["links"] if l["lhs"][0] != 8]
    y2 = graph_eval.run_graph(G.Graph(json.dumps(doc)), x)
    assert np.allclose(y * O.link_divisor(3), y2 * O.link_divisor(2), rtol=1e-6, atol=1e-7)


def test_mux_demux_routing_semantics(dspfx):
    from dsp_stuff_amd import graph as G
    x = O.noise(9, np.arange(2), np.arange(256))
    hop = lambda v: (v / O.link_divisor(1)).astype(F)
    for in_port, out_port in (("A", "A"), ("B", "A"), ("A", "B"), ("B", "B")):
        g = G.Graph(graphs.routing(in_port, out_port))
        y = graph_eval.run_graph(g, x)
        # by hand: branch -> mux (identity after its hop) -> demux (identity after its hop) -> effect -> output average of 2 pipes
        branch = O.run_channels([(dspfx.Gain(0.5) if in_port == "A" else dspfx.LowPass(0.4)).oracle_desc()], x, 3)
        routed = hop(hop(branch))
        eff = dspfx.Distort(2.0, dspfx.HARD_CLIP) if out_port == "A" else dspfx.BiQuad(**graphs.BQ)
        e = O.run_channels([eff.oracle_desc()], routed, 3)
        other = O.run_channels([(dspfx.BiQuad(**graphs.BQ) if out_port == "A" else dspfx.Distort(2.0, dspfx.HARD_CLIP)).oracle_desc()],
                               np.zeros_like(x), 3)
        pipes = (e, other) if out_port == "A" else (other, e)
        want = ((F(0) + pipes[0] + pipes[1]) / O.link_divisor(2)).astype(F)
        assert np.array_equal(y, want), (in_port, out_port)
        runs, _ = G.plan_runs(g)
        # the selected branch > mux > demux > the selected effect fuse into one run; the unread branch and the
        # effect that runs on zeros are runs of their own
        got = sorted(sorted(m.id for m in r.nodes) for r in runs)
        branch_id, unread = (1, 2) if in_port == "A" else (2, 1)
        sel, other_id = (5, 6) if out_port == "A" else (6, 5)
        assert got == sorted([sorted([branch_id, 3, 4, sel]), [unread], [other_id]]), got


def test_display_only_nodes_are_dropped(dspfx):
    """pitch / wave_view / spectrogram write no output (pitch.rs:120-146, wave_view.rs:157-175): a graph with
    them evaluates like the graph without them, and the producer they tap keeps fusing with its only real consumer."""
    from dsp_stuff_amd import config, graph as G
    chain = [dspfx.BiQuad(), dspfx.Gain(0.5), dspfx.HighPass(0.2)]
    doc = json.loads(config.dump_dspconfig(chain))
    tap = doc["nodes"][1]                                   # the biquad
    for k, tn in enumerate(("wave_view", "pitch", "spectrogram")):
        nid, pid = 500 + k, 600 + k
        doc["nodes"].append({"id": nid, "typename": tn, "position": [0, 0], "cfg": {"id": nid, "inputs": {"in": pid}, "outputs": {}}})
        doc["links"].append({"lhs": [tap["id"], tap["cfg"]["outputs"]["out"]], "rhs": [nid, pid]})
    g, plain = G.Graph(json.dumps(doc)), G.Graph(config.dump_dspconfig(chain))
    assert sorted(g.dropped) == [500, 501, 502] and set(g.nodes) == set(plain.nodes)
    runs, _ = G.plan_runs(g)
    assert len(runs) == 1 and len(runs[0].nodes) == 3
    x = O.noise(4, np.arange(2), np.arange(256))
    assert np.array_equal(graph_eval.run_graph(g, x), graph_eval.run_graph(plain, x))
    bad = json.loads(config.dump_dspconfig(chain))
    bad["nodes"][1]["typename"] = "muff"
    try:
        G.Graph(json.dumps(bad))
        assert False
    except config.DspConfigError as e:
        assert "outside the accelerated path" in str(e)


def test_fused_plan_is_the_graph_in_link_order(dspfx):
    """What `dspfx_graph_set` receives: nodes in topological order, every link forward, each port's links in the
    document's order (the order collect_and_average adds them in)."""
    from dsp_stuff_amd import graph as G
    E = dspfx
    specs, links = G.fused_plan(G.Graph(graphs.diamond()))
    assert [s.kind for s in specs] == [E.GAIN, E.BIQUAD, E.HIGH_PASS, E.ADD, E.DISTORT, E.REVERB]
    assert links == [(E.GRAPH_INPUT, 0, E.PORT_MAIN), (E.GRAPH_INPUT, 1, E.PORT_MAIN), (1, 2, E.PORT_MAIN),
                     (0, 3, E.PORT_MAIN), (2, 3, E.PORT_SIDE), (3, 4, E.PORT_MAIN), (0, 4, E.PORT_MAIN),
                     (4, 5, E.PORT_MAIN), (5, 6, E.PORT_MAIN), (2, 6, E.PORT_MAIN)]
    specs, links = G.fused_plan(G.Graph(graphs.lfo_tremolo()))
    assert (0, 3, E.PORT_SLIDER + 0) in links and (1, 4, E.PORT_SLIDER + 0) in links   # LFO -> gain.level, envelope -> mix.ratio
    specs, links = G.fused_plan(G.Graph(graphs.routing("B", "A")))
    assert any(s == E.GRAPH_ZERO for s, _, _ in links)          # the demux's unselected output: a connected pipe of zeros
    for seed in range(30):
        n = 8 if seed < 20 else 16
        g = G.Graph(graphs.random_dag(seed, n, libm=bool(seed % 2)))
        specs, links = G.fused_plan(g)
        assert len(specs) == n and all(s < d for s, d, _ in links) and all(0 <= d <= n for _, d, _ in links)
    assert G.fused_plan(G.Graph(graphs.random_dag(1, 17))) is None            # more nodes than one kernel holds
    fuzz = json.loads(graphs.diamond())
    next(n for n in fuzz["nodes"] if n["id"] == 5)["cfg"]["mode"] = "Fuzz"
    assert G.fused_plan(G.Graph(json.dumps(fuzz))) is None                   # Fuzz is block-global: its own kernel


def _compile_generated(dspfx, specs, links, tmp_path, tag):
    """hipcc the translation unit the engine would hand to hiprtc, with the kernel instantiated both ways."""
    import os
    import subprocess
    src = dspfx.graph_source(specs, links)
    assert "struct Prog" in src and '#include "graph_kernel.hip.h"' in src
    f = tmp_path / f"{tag}.hip"
    f.write_text(src + "template __global__ void dspfx::graph_kernel<8, 2, dspfx::Prog>(const dspfx::GraphArgs);\n"
                       "template __global__ void dspfx::graph_kernel<8, 1, dspfx::Prog>(const dspfx::GraphArgs);\n")
    csrc = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "dsp-stuff_amd", "csrc")
    r = subprocess.run(["hipcc", "--offload-arch=gfx950", "-O1", "-std=c++17", "-ffp-contract=off", "-fno-fast-math", f"-I{csrc}",
                        "-c", str(f), "-o", str(tmp_path / f"{tag}.o")], capture_output=True, text=True)
    assert r.returncode == 0, src + "\n" + r.stderr
    return src


def test_generated_graph_kernels_compile_without_a_gpu(dspfx, tmp_path):
    """The code generator behind dspfx_graph_set, exercised where there is no device: every port form it emits
    (fan-in with the divisor spelled out, pipes of zeros, "b" ports, slider ports fed by nodes, unplugged ports,
    generators, delay-tap prefetch, a slot beyond the chain argument block) must be valid against graph_kernel.hip.h."""
    from dsp_stuff_amd import graph as G
    E = dspfx
    src = _compile_generated(E, *G.fused_plan(G.Graph(graphs.diamond())), tmp_path, "diamond")
    assert "g_acc<F, CPL>(v3, v0)" in src and "g_add<F, CPL>(v3, b3)" in src           # Add: main <- gain, "b" <- high_pass
    assert "ring_prefetch<F, CPL, false>(gslot<5>(g)" in src and src.count("g_div<") == 8   # one division per connected port
    assert "0x1.00034" in src                                                           # f32(0.0001 + 2): two links into a port
    src = _compile_generated(E, *G.fused_plan(G.Graph(graphs.lfo_tremolo())), tmp_path, "lfo")
    assert "gain_mod_core<F, CPL>(v3, p3_0)" in src and "g_mix_mod<F, CPL>(v4, b4, p4_0)" in src
    src = _compile_generated(E, *G.fused_plan(G.Graph(graphs.routing("B", "A"))), tmp_path, "routing")
    assert "g_acc_zero<F, CPL>" in src
    _compile_generated(E, *G.fused_plan(G.Graph(graphs.random_dag(116, 16, libm=True))), tmp_path, "random16")
    with pytest.raises(E.DspfxError):
        E.graph_source([E.Gain(1.0)], [(0, 0, E.PORT_MAIN)])                           # a link must go forward


def test_series_plan_cuts_at_the_nodes_all_signal_passes(dspfx):
    from dsp_stuff_amd import graph as G
    E = dspfx
    g = G.Graph(graphs.cab_rig())
    assert G.fused_plan(g) is None                                   # a FIR node is its own kernel
    steps = G.series_plan(g)
    assert [s[0] for s in steps] == ["graph", "node", "graph"]
    specs, links = steps[0][1], steps[0][2]
    assert steps[0][3:] == (-1, None) and steps[1][2] == 0 and steps[2][3:] == (1, None)   # whose block each step reads
    assert [s.kind for s in specs] == [E.GAIN, E.DISTORT, E.BIQUAD, E.ADD]
    assert links[-2:] == [(3, 4, E.PORT_MAIN), (1, 4, E.PORT_MAIN)]  # the FIR node's two incoming links = the segment's Output
    assert steps[1][1].kind == E.FIR
    specs, links = steps[2][1], steps[2][2]
    assert [s.kind for s in specs] == [E.REVERB, E.MIX, E.HIGH_PASS]
    assert links[:3] == [(E.GRAPH_INPUT, 0, E.PORT_MAIN), (E.GRAPH_INPUT, 1, E.PORT_MAIN), (0, 1, E.PORT_SIDE)]
    assert G.series_plan(G.Graph(graphs.cab_rig(bypass=True))) is None   # a link around the FIR node next to others into it
    # wet / dry: ONE signal feeds the FIR node and also goes on beside it
    steps = G.series_plan(G.Graph(graphs.cab_rig(dry=True)))
    assert [s[0] for s in steps] == ["graph", "node_hop", "graph"]
    assert steps[0][2][-1] == (3, 4, E.PORT_MAIN | E.PORT_RAW)          # the Add's output handed over as it is
    assert (E.GRAPH_INPUT2, 1, E.PORT_SIDE) in steps[2][2]              # ... and read as the Mix's "b" port after the FIR node
    assert steps[1][2] == 0 and steps[2][3:] == (1, 0)                  # the FIR reads step 0; the last kernel steps 1 and 0
    # wet / dry over a whole long rig: the graph's Input block stays alive beside every stage, as each kernel's second block
    for seed in range(6):
        g = G.Graph(graphs.long_rig(seed, 12, fir_at=5 if seed % 2 else None, dry_mix=True))
        steps = G.series_plan(g)
        assert steps is not None and len(steps) >= 2, seed
        last = steps[-1]
        assert last[0] == "graph" and last[4] == -1 and any(l[0] == E.GRAPH_INPUT2 for l in last[2])
        for st in steps[1:]:
            if st[0] == "graph":
                assert st[4] == -1                                       # carried from the very first block
    assert G.series_plan(G.Graph(graphs.diamond())) is None              # nothing to cut at


def test_segment_plan_cuts_long_graphs_where_one_signal_crosses(dspfx, tmp_path):
    from dsp_stuff_amd import graph as G
    E = dspfx
    for seed in range(12):
        g = G.Graph(graphs.long_rig(seed, 12, fir_at=5 if seed % 2 else None))
        n_nodes = sum(1 for n in g.nodes.values() if n.spec is not None)
        steps = G.series_plan(g)
        assert steps is not None and n_nodes > E.GRAPH_MAX_NODES
        assert sum(len(s[1]) if s[0] == "graph" else 1 for s in steps) == n_nodes          # every node exactly once
        for k, (kind, *what) in enumerate(steps):
            if kind != "graph":
                continue
            specs, links = what[:2]
            assert len(specs) <= E.GRAPH_MAX_NODES and all(s < d for s, d, _ in links)
            raw_out = [l for l in links if l[2] & E.PORT_RAW]
            hands_over = k + 1 < len(steps) and steps[k + 1][0] == "graph"               # next is a kernel, not a FIR node
            assert len(raw_out) == (1 if hands_over else 0) and all(l[1] == len(specs) for l in raw_out)
    # the RAW handover in generated code: a copy, not an average
    g = G.Graph(graphs.long_rig(0, 12))
    specs, links = G.series_plan(g)[0][1:3]
    src = _compile_generated(E, specs, links, tmp_path, "handover")
    assert "g_copy<F, CPL>(ys[0], v%d)" % (len(specs) - 1) in src
    # a kernel of at most 3 nodes: the diamond is cut twice, each time with an older signal carried beside the new one
    steps = G.segment_plan(G.Graph(graphs.diamond()), max_nodes=3)
    assert [(s[0], len(s[1]), s[3], s[4]) for s in steps] == [("graph", 1, -1, None), ("graph", 2, 0, -1), ("graph", 3, 1, 0)]
    assert G.segment_plan(G.Graph(graphs.random_dag(3, 12)), max_nodes=4) is None    # wide fan-in everywhere: nothing to cut at


def _graph_goldens():
    import glob
    import os
    d = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "graphs")
    out = []
    for path in sorted(glob.glob(os.path.join(d, "*.npz"))):
        z = np.load(path)
        out.append((os.path.basename(path)[:-4], str(z["doc"]), z["x"], z["y"]))
    return out


def test_graph_golden_vectors_pin_the_oracle(dspfx):
    """tests/golden/graphs/*.npz (made by tests/golden/make_graph_golden.py): the committed documents still parse to the
    same graphs and the oracle's node-by-node evaluation still gives the committed outputs, bit for bit."""
    from dsp_stuff_amd import graph as G
    cases = _graph_goldens()
    assert len(cases) >= 8
    for name, doc, x, y in cases:
        got = graph_eval.run_graph(G.Graph(doc), x)
        assert np.array_equal(got.view(np.uint32), y.view(np.uint32)), name


def test_region_plan_covers_any_graph(dspfx, tmp_path):
    """region_plan: ANY graph as a few generated kernels with several input / output blocks each.  Every node lands in
    exactly one step, links inside a region go forward, a region reads / writes at most GRAPH_MAX_IO blocks, every ref
    points at an earlier step, and the generated multi-block source compiles without a GPU."""
    from dsp_stuff_amd import graph as G
    E = dspfx
    for seed, n in [(1, 40), (2, 40), (3, 30), (6, 40), (11, 23), (12, 17), (13, 5)]:
        g = G.Graph(graphs.random_dag(seed, n))
        steps = G.region_plan(g)
        assert steps is not None
        n_nodes = sum(1 for m in g.nodes.values() if m.spec is not None)
        assert sum(len(s[1]) if s[0] == "region" else (1 if s[0] == "node" else 0) for s in steps) == n_nodes
        assert sum(1 for s in steps if s[0] == "region") <= -(-n_nodes // E.GRAPH_MAX_NODES) + 1
        for k, (kind, *what) in enumerate(steps):
            if kind != "region":
                continue
            specs, links, in_refs, n_out = what
            assert len(specs) <= E.GRAPH_MAX_NODES and len(in_refs) <= E.GRAPH_MAX_IO and 1 <= n_out <= E.GRAPH_MAX_IO
            assert all(r == -1 or (r[0] < k and r[1] < (steps[r[0]][4] if steps[r[0]][0] == "region" else 1)) for r in in_refs)
            for s_, d_, p_ in links:
                assert s_ < d_ or s_ < 0
                assert d_ < len(specs) + n_out
                if s_ in E.GRAPH_INPUTS:
                    assert E.GRAPH_INPUTS.index(s_) < len(in_refs)
            written = {d_ - len(specs) for _, d_, _ in links if d_ >= len(specs)}
            assert written == set(range(n_out)) or (0 not in written and k + 1 == len(steps))   # an Output node with nothing plugged in
    # FIR / Fuzz nodes stay steps of their own, with any number of signals around them and a node-fed level port
    g = G.Graph(graphs.around_fir_and_fuzz())
    assert G.fused_plan(g) is None and G.series_plan(g) is None
    steps = G.region_plan(g)
    kinds = [s[0] for s in steps]
    assert kinds.count("node") == 2 and kinds[-1] == "region"
    fuzz = next(s for s in steps if s[0] == "node" and s[1].kind == E.DISTORT)
    assert list(fuzz[3]) == [0] and fuzz[3][0] is not None                      # its level slider reads a block
    # the multi-block source compiles: several inputs, several outputs
    g = G.Graph(graphs.random_dag(2, 40))
    steps = G.region_plan(g)
    mid = steps[1]
    assert len(mid[3]) >= 3 and mid[4] >= 2
    src = _compile_generated(E, mid[1], mid[2], tmp_path, "region")
    assert "xs[2]" in src and "ys[1]" in src and "static constexpr int n_out = %d" % mid[4] in src


def test_output_blocks_must_be_contiguous(dspfx):
    """A generated kernel stores every output block below the highest one a link names; a graph that writes block 2 but
    not block 1 would store through the NULL entry the header allows for unused blocks.  Rejected when the graph is
    validated (dspfx_graph_set / dspfx_graph_source), no device needed."""
    nodes = [dspfx.Gain(0.5), dspfx.LowPass(0.3)]
    ok = [(dspfx.GRAPH_INPUT, 0, dspfx.PORT_MAIN), (0, 1, dspfx.PORT_MAIN), (1, 2, dspfx.PORT_MAIN), (0, 3, dspfx.PORT_MAIN)]
    assert "n_out = 2" in dspfx.graph_source(nodes, ok)
    gap = [(dspfx.GRAPH_INPUT, 0, dspfx.PORT_MAIN), (0, 1, dspfx.PORT_MAIN), (1, 2, dspfx.PORT_MAIN), (0, 4, dspfx.PORT_MAIN)]
    with pytest.raises(dspfx.DspfxError) as ei:
        dspfx.graph_source(nodes, gap)
    assert ei.value.status == -1
