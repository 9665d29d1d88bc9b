"""Whole saved graphs (DAGs) on the GPU against the reference-semantics evaluation on the CPU oracle."""
import numpy as np
import pytest

import graph_eval
import graphs
import oracle as O
from chains import ulp_diff

pytestmark = pytest.mark.gpu
F = np.float32


@pytest.fixture(scope="module")
def G(dspfx):
    from dsp_stuff_amd import graph
    return graph


NAMES = ["diamond", "lfo_tremolo", "fan_in_three", "routing", "routing_ab", "routing_ba"]


def graph_text(name):
    return {"routing_ab": lambda: graphs.routing("A", "B"), "routing_ba": lambda: graphs.routing("B", "A")}.get(
        name, getattr(graphs, name, None))()


@pytest.mark.parametrize("fused", [None, False])
@pytest.mark.parametrize("name", NAMES)
@pytest.mark.parametrize("N,tile,B", [(100, 0, 128), (128, 64, 128), (64, 0, 256)])
def test_graph_matches_reference_semantics(dspfx, G, name, N, tile, B, fused):
    """fused=None: the whole graph as one generated kernel (N = 100 is not a whole number of waves and falls back to
    run-by-run evaluation); fused=False: always run by run."""
    import torch
    text = graph_text(name)
    nf = 768
    x = O.noise(0x5EED0001, np.arange(N), np.arange(nf))
    ge = G.GraphEngine(text, N, B, tile_channels=tile, fused=fused)
    assert (ge.fused is not None) == (fused is None and N % 64 == 0), ge.describe()
    got = np.empty_like(x)
    for f0 in range(0, nf, B):
        dx = torch.from_numpy(dspfx.to_layout(x[f0:f0 + B], tile)).cuda()
        y = ge.process(dx, B)
        torch.cuda.synchronize()
        got[f0:f0 + B] = dspfx.from_layout(y.cpu().numpy(), B, N, tile)
    ref = graph_eval.run_graph(ge.g, x)
    d = ulp_diff(got, ref)
    assert d.max() <= 1, (name, N, tile, B, int(d.max()), ge.describe())
    ge.close()


@pytest.mark.parametrize("name", NAMES)
@pytest.mark.parametrize("N,tile", [(229376, 256), (132096, 256), (4096, 0)])
def test_one_kernel_graph_equals_run_by_run(dspfx, G, name, N, tile):
    """The generated whole-graph kernel and the run-by-run evaluation perform the same f32 operations in the same
    order: bit-identical outputs, block after block (two channels per lane from 229376 tiled channels on)."""
    import torch
    B, blocks = 128, 4
    a = G.GraphEngine(graph_text(name), N, B, tile_channels=tile, fused=True)
    b = G.GraphEngine(graph_text(name), N, B, tile_channels=tile, fused=False)
    assert "jit_graph" in a.describe(), a.describe()
    if N >= 229376:
        assert "_c2" in a.describe(), a.describe()
    for k in range(blocks):
        x = torch.empty(B * N, dtype=torch.float32, device="cuda")
        a.util.fill_noise(x, B, k * B, 0x5EED0002)
        ya = a.process(x, B).clone()
        yb = b.process(x, B)
        torch.cuda.synchronize()
        assert torch.equal(ya.view(torch.int32), yb.view(torch.int32)), (name, N, tile, k)
    a.close()
    b.close()


@pytest.mark.parametrize("seed", range(24))
def test_random_dag_matches_reference_semantics(dspfx, G, seed):
    """Random DAGs of 8 (seeds < 16) or 16 nodes (fan-in up to 3 on every port, fan-out, unplugged ports, slider ports
    fed by nodes) of the exact-arithmetic kinds, as one generated kernel, against the node-by-node evaluation of the
    oracle."""
    import torch
    N, B, nf = 64, 128, 512
    tile = 64 if seed % 2 else 0
    text = graphs.random_dag(seed, 8 if seed < 16 else 16)
    x = O.noise(0x5EED0003 + seed, np.arange(N), np.arange(nf))
    ge = G.GraphEngine(text, N, B, tile_channels=tile, fused=True)
    got = np.empty_like(x)
    for f0 in range(0, nf, B):
        y = ge.process(torch.from_numpy(dspfx.to_layout(x[f0:f0 + B], tile)).cuda(), B)
        torch.cuda.synchronize()
        got[f0:f0 + B] = dspfx.from_layout(y.cpu().numpy(), B, N, tile)
    ref = graph_eval.run_graph(ge.g, x)
    assert np.isfinite(ref).all()
    d = ulp_diff(got, ref)
    assert d.max() <= 1, (seed, int(d.max()), text)
    assert np.array_equal(np.signbit(got), np.signbit(ref)), seed          # +0 / -0 included (ulp_diff calls them equal)
    ge.close()


@pytest.mark.parametrize("seed", range(100, 124))
def test_random_dag_one_kernel_equals_run_by_run(dspfx, G, seed):
    """Random DAGs (8 nodes, 16 from seed 116) of every fusable kind (libm nodes included): the generated kernel and
    the run-by-run evaluation are bit-identical."""
    import torch
    N, B = 4096, 128
    text = graphs.random_dag(seed, 8 if seed < 116 else 16, libm=True)
    a = G.GraphEngine(text, N, B, fused=True)
    b = G.GraphEngine(text, N, B, fused=False)
    for k in range(4):
        x = torch.empty(B * N, dtype=torch.float32, device="cuda")
        a.util.fill_noise(x, B, k * B, 0x5EED0004 + seed)
        ya = a.process(x, B).clone()
        yb = b.process(x, B)
        torch.cuda.synchronize()
        assert torch.equal(ya.view(torch.int32), yb.view(torch.int32)), (seed, k, text)
    a.close()
    b.close()


@pytest.mark.parametrize("name", ["diamond", "lfo_tremolo", "fan_in_three"])
@pytest.mark.parametrize("nf", [100, 37, 1])
def test_one_kernel_graph_short_blocks(dspfx, G, name, nf):
    """Blocks that are not a multiple of the kernel's 8-frame chunk (the single-frame tail loop) and shorter than the
    reference's 128: still bit-identical to the run-by-run evaluation, block after block."""
    import torch
    N, B = 1024, 128
    a = G.GraphEngine(graph_text(name), N, B, fused=True)
    b = G.GraphEngine(graph_text(name), N, B, fused=False)
    x = torch.empty(B * N, dtype=torch.float32, device="cuda")
    for k in range(5):
        a.util.fill_noise(x, nf, k * nf, 0x5EED0007)
        ya = a.process(x, nf).clone()
        yb = b.process(x, nf)
        torch.cuda.synchronize()
        assert torch.equal(ya[:nf * N].view(torch.int32), yb[:nf * N].view(torch.int32)), (name, nf, k)
    a.close()
    b.close()


def test_graph_engine_mix_bus_host_path_and_parameter_changes(dspfx, G):
    """A graph engine is an engine: the mix bus, the host-buffer path and slider / mode changes work as on a chain."""
    import json
    import torch
    E = dspfx
    N, B = 2048, 128
    doc = graphs.diamond()
    specs, links = G.fused_plan(G.Graph(doc))
    eng = E.Engine(N, B)
    eng.set_graph(specs, links)
    x = torch.empty(B * N, dtype=torch.float32, device="cuda")
    eng.fill_noise(x, B, 0, 0x5EED0005)
    y = torch.empty_like(x)
    mix = torch.empty(B, dtype=torch.float32, device="cuda")
    eng.process(x, out=y, mix=mix, n_frames=B)
    torch.cuda.synchronize()
    yh = y.cpu().numpy().reshape(B, N)
    assert np.allclose(mix.cpu().numpy(), yh.astype(np.float64).sum(axis=1), rtol=1e-5, atol=1e-3)
    # algorithmic traffic of the graph kernel: one read + one write + the delay node's tap and store + filter state
    assert abs(eng.algorithmic_bytes_per_sample(B) - (8 + 8 + (32 + 8) / B)) < 1e-9          # biquad 32/B, high_pass 8/B; "b" ports cost nothing
    # host-buffer path of a second engine: same block, same bits
    eng2 = E.Engine(N, B)
    eng2.set_graph(specs, links)
    out_h = eng2.process_host(x.cpu().numpy().reshape(B, N))
    assert np.array_equal(out_h.view(np.uint32), yh.view(np.uint32))
    eng2.close()
    # slider and mode changes on the fused graph == the same graph saved with those settings
    eng.set_param(0, 0, 0.25)                   # gain.level
    eng.set_mode(4, E.HARD_CLIP)                # distort: SoftClip -> HardClip (a different generated kernel)
    eng.reset()
    mod = json.loads(doc)
    next(n for n in mod["nodes"] if n["id"] == 1)["cfg"]["level"] = 0.25
    next(n for n in mod["nodes"] if n["id"] == 5)["cfg"]["mode"] = "HardClip"
    ref = G.GraphEngine(json.dumps(mod), N, B, fused=False)
    for k in range(3):
        eng.fill_noise(x, B, k * B, 0x5EED0006)
        eng.process(x, out=y, n_frames=B)
        yr = ref.process(x, B)
        torch.cuda.synchronize()
        assert torch.equal(y.view(torch.int32), yr.view(torch.int32)), k
    with pytest.raises(E.DspfxError):           # control ports of a fused graph are links, not call arguments
        eng.process(x, out=y, n_frames=B, ctl={(0, 0): x})
    eng.set_chain([E.Gain(0.5)])                # back to a chain engine
    assert "jit_graph" not in eng.describe()
    ref.close()
    eng.close()


def test_long_chain_as_one_graph_kernel(dspfx, monkeypatch):
    """A 12-node chain is two launches for dspfx_chain_set (8 + 4 nodes) and one for dspfx_graph_set; the results
    differ only by the Output node's hop, which the graph form includes."""
    import torch
    E = dspfx
    N, B = 2048, 128
    chain = [E.Gain(0.9), E.BiQuad(1.0, -1.2, 0.5, 0.3, 0.2, 0.1), E.Distort(3.0, E.SOFT_CLIP), E.LowPass(0.3),
             E.Reverb(delay_samples=256, decay=0.4), E.HighPass(0.2), E.Gain(1.1), E.Distort(2.0, E.HARD_CLIP),
             E.BiQuad(1.0, -0.5, 0.2, 0.4, 0.1, 0.0), E.Envelope(4.0, 100.0), E.Reverb(delay_samples=384, decay=0.3), E.Gain(0.7)]
    monkeypatch.setenv("DSPFX_JIT", "0")       # the chain engine proper: interpreter, 8 + 4 nodes
    a = E.Engine(N, B, link_flags=E.LINK_INTERNAL | E.LINK_INPUT)
    a.set_chain(chain)
    assert a.describe().count("stage") >= 2
    g = E.Engine(N, B)
    g.set_graph(chain, [(E.GRAPH_INPUT, 0, E.PORT_MAIN)] + [(i, i + 1, E.PORT_MAIN) for i in range(len(chain))])
    assert g.describe().count("fused kernel") == 1 and "jit_graph" in g.describe()
    x = torch.empty(B * N, dtype=torch.float32, device="cuda")
    ya, yg = torch.empty_like(x), torch.empty_like(x)
    hop = np.float32(O.link_divisor(1))
    for k in range(4):
        a.fill_noise(x, B, k * B, 0x5EED0008)
        a.process(x, out=ya, n_frames=B)
        g.process(x, out=yg, n_frames=B)
        torch.cuda.synchronize()
        want = ((np.float32(0.0) + ya.cpu().numpy()) / hop).astype(np.float32)
        assert np.array_equal(yg.cpu().numpy().view(np.uint32), want.view(np.uint32)), k
    a.close()
    g.close()


@pytest.mark.parametrize("N,tile", [(128, 0), (256, 64)])
def test_graph_with_a_fir_node_in_series(dspfx, G, N, tile):
    """An amp stage, a cabinet impulse response (FIR: its own MFMA kernel) and a delay tail: the graph is cut at the FIR
    node and each side is one generated kernel.  Bit-identical to the run-by-run evaluation (same FIR kernel, same
    inputs); against the oracle within the FIR path's tolerance.  A link around the FIR node makes it run by run."""
    import torch
    B, nf = 128, 768
    text = graphs.cab_rig()
    x = O.noise(0x5EED000C, np.arange(N), np.arange(nf))
    a = G.GraphEngine(text, N, B, tile_channels=tile)
    b = G.GraphEngine(text, N, B, tile_channels=tile, fused=False)
    assert len(a.series) == 3 and a.describe().count("jit_graph") == 2 and "fir kernel" in a.describe(), a.describe()
    got = np.empty_like(x)
    for f0 in range(0, nf, B):
        dx = torch.from_numpy(dspfx.to_layout(x[f0:f0 + B], tile)).cuda()
        ya = a.process(dx, B).clone()
        yb = b.process(dx, B)
        torch.cuda.synchronize()
        assert torch.equal(ya.view(torch.int32), yb.view(torch.int32)), f0
        got[f0:f0 + B] = dspfx.from_layout(ya.cpu().numpy(), B, N, tile)
    ref = graph_eval.run_graph(a.g, x)
    assert np.abs(got - ref).max() <= 1e-5 * np.abs(ref).max()
    a.close()
    b.close()
    # a link AROUND the cabinet next to others into it: no series form; the general region plan takes it (the FIR node
    # between two generated kernels, the bypassing signal one of the blocks they exchange), run by run only on request
    c = G.GraphEngine(graphs.cab_rig(bypass=True), N, B, tile_channels=tile)
    assert not c.series and c.fused is None and c.regions and [st["kind"] for st in c.regions].count("node") == 1
    c.close()
    c = G.GraphEngine(graphs.cab_rig(bypass=True), N, B, tile_channels=tile, fused=False)
    assert not c.series and not c.regions and len(c.runs) > 1
    c.close()
    # wet / dry: the signal that feeds the cabinet also goes on beside it -> handed over raw, the FIR engine applies its own
    # hop, the kernel after it reads the FIR output as Input and the dry signal as its second block
    a = G.GraphEngine(graphs.cab_rig(dry=True), N, B, tile_channels=tile)
    b = G.GraphEngine(graphs.cab_rig(dry=True), N, B, tile_channels=tile, fused=False)
    assert a.series_kind == [("graph", -1, None), ("node_hop", 0, None), ("graph", 1, 0)], a.series_kind
    got = np.empty_like(x)
    for f0 in range(0, nf, B):
        dx = torch.from_numpy(dspfx.to_layout(x[f0:f0 + B], tile)).cuda()
        ya = a.process(dx, B).clone()
        yb = b.process(dx, B)
        torch.cuda.synchronize()
        assert torch.equal(ya.view(torch.int32), yb.view(torch.int32)), f0
        got[f0:f0 + B] = dspfx.from_layout(ya.cpu().numpy(), B, N, tile)
    ref = graph_eval.run_graph(a.g, x)
    assert np.abs(got - ref).max() <= 1e-5 * np.abs(ref).max()
    with pytest.raises(dspfx.DspfxError):            # the second block is not optional for that kernel
        a.series[2][0].process(torch.from_numpy(dspfx.to_layout(x[:B], tile)).cuda(), n_frames=B)
    a.close()
    b.close()
    # the same rig around a Fuzz node (block-global over 128 frames: its own kernel too)
    a = G.GraphEngine(graphs.cab_rig(cut="fuzz"), N, B, tile_channels=tile)
    b = G.GraphEngine(graphs.cab_rig(cut="fuzz"), N, B, tile_channels=tile, fused=False)
    assert len(a.series) == 3 and "fuzz kernel" in a.describe(), a.describe()
    for f0 in range(0, nf, B):
        dx = torch.from_numpy(dspfx.to_layout(x[f0:f0 + B], tile)).cuda()
        ya = a.process(dx, B).clone()
        yb = b.process(dx, B)
        torch.cuda.synchronize()
        assert torch.equal(ya.view(torch.int32), yb.view(torch.int32)), f0
    a.close()
    b.close()


@pytest.mark.parametrize("seed", range(8))
def test_long_graph_as_a_series_of_kernels(dspfx, G, seed):
    """25-32 node pedalboards (stages in series, each a small sub-graph): more than one kernel holds, so the graph is
    cut where a single signal crosses (the handover link is RAW) and, on odd seeds, around a FIR node.  Bit-identical
    to the run-by-run evaluation; <= 1 ulp against the oracle (FIR tolerance on odd seeds)."""
    import torch
    N, B, nf = 128, 128, 512
    tile = 64 if seed % 4 >= 2 else 0
    text = graphs.long_rig(seed, 12, fir_at=5 if seed % 2 else None)
    a = G.GraphEngine(text, N, B, tile_channels=tile)
    b = G.GraphEngine(text, N, B, tile_channels=tile, fused=False)
    assert a.fused is None and len(a.series) >= 2 and len(a.series) < len(b.runs), a.describe()
    x = O.noise(0x5EED000D + seed, np.arange(N), np.arange(nf))
    got = np.empty_like(x)
    for f0 in range(0, nf, B):
        dx = torch.from_numpy(dspfx.to_layout(x[f0:f0 + B], tile)).cuda()
        ya = a.process(dx, B).clone()
        yb = b.process(dx, B)
        torch.cuda.synchronize()
        assert torch.equal(ya.view(torch.int32), yb.view(torch.int32)), (seed, f0)
        got[f0:f0 + B] = dspfx.from_layout(ya.cpu().numpy(), B, N, tile)
    ref = graph_eval.run_graph(a.g, x[:, :16])
    assert np.isfinite(ref).all() and np.abs(ref).max() > 0
    if seed % 2:
        assert np.abs(got[:, :16] - ref).max() <= 1e-5 * np.abs(ref).max()
    else:
        assert ulp_diff(got[:, :16], ref).max() <= 1, seed
    a.close()
    b.close()


def test_graph_golden_vectors(dspfx, G):
    """The HIP path against the committed graph golden vectors (tests/golden/graphs/): every plan form -- one kernel,
    segments around a FIR node (with and without a dry path), a long graph in two kernels."""
    import torch
    from test_graph_cpu import _graph_goldens
    forms = set()
    for name, doc, x, y in _graph_goldens():
        N, B = 64, 128
        xs = np.tile(x, (1, 22))[:, :N]                       # the 3 channels repeated across a wave
        ge = G.GraphEngine(doc, N, B)
        forms.add("one" if ge.fused is not None else "segments" if ge.series else "runs")
        got = np.empty_like(xs)
        for f0 in range(0, xs.shape[0], B):
            out = ge.process(torch.from_numpy(xs[f0:f0 + B].copy()).cuda(), B)
            torch.cuda.synchronize()
            got[f0:f0 + B] = out.cpu().numpy().reshape(B, N)
        assert np.array_equal(got[:, :3], got[:, 3:6]), name          # channels are independent
        if "fir" in name or "cab" in name:
            assert np.abs(got[:, :3] - y).max() <= 1e-5 * np.abs(y).max(), name
        else:
            assert ulp_diff(got[:, :3], y).max() <= 1, name
        ge.close()
    assert forms == {"one", "segments"}, forms


@pytest.mark.parametrize("max_nodes", [2, 5])
def test_cutting_into_small_kernels_changes_nothing(dspfx, G, max_nodes):
    """The series planner under stress: pretend a kernel holds only 2 / 3 / 5 nodes, so that graphs which are normally one
    kernel get cut wherever one new signal crosses (with an older one carried beside it as the second block).  Whatever
    can be cut must give the one-kernel result bit for bit."""
    import torch
    N, B = 256, 128
    cut = 0
    docs = [graph_text(n) for n in NAMES] + [graphs.long_rig(s, 4, dry_mix=bool(s % 2)) for s in range(8)] + \
           [graphs.random_dag(s, 8) for s in range(8)]
    for k, text in enumerate(docs):
        steps = G.segment_plan(G.Graph(text), max_nodes)
        if steps is None or len(steps) < 2:
            continue
        cut += 1
        a = G.GraphEngine(text, N, B, max_nodes=max_nodes)
        b = G.GraphEngine(text, N, B, fused=True)
        assert len(a.series) >= 2, (k, a.describe())
        x = torch.empty(B * N, dtype=torch.float32, device="cuda")
        for blk in range(4):
            b.util.fill_noise(x, B, blk * B, 0x5EED000E + k)
            ya = a.process(x, B).clone()
            yb = b.process(x, B)
            torch.cuda.synchronize()
            assert torch.equal(ya.view(torch.int32), yb.view(torch.int32)), (k, max_nodes, blk, a.series_kind)
        a.close()
        b.close()
    assert cut >= 6, cut


@pytest.mark.parametrize("seed", range(6))
def test_wet_dry_over_a_long_rig(dspfx, G, seed):
    """The graph's Input mixed back in after 25-32 nodes (and a FIR node on odd seeds): the Input block stays alive
    beside every kernel of the series as its second block."""
    import torch
    N, B, nf = 128, 128, 512
    text = graphs.long_rig(seed, 12, fir_at=5 if seed % 2 else None, dry_mix=True)
    a = G.GraphEngine(text, N, B)
    b = G.GraphEngine(text, N, B, fused=False)
    assert len(a.series) >= 2 and a.series_kind[-1][2] == -1, a.series_kind
    a.tune_placement(torch.zeros(B * N, dtype=torch.float32, device="cuda"))     # a no-op at this size, but every step's buffers resolve
    x = O.noise(0x5EED000F + seed, np.arange(N), np.arange(nf))
    got = np.empty_like(x)
    for f0 in range(0, nf, B):
        dx = torch.from_numpy(x[f0:f0 + B].copy()).cuda()
        ya = a.process(dx, B).clone()
        yb = b.process(dx, B)
        torch.cuda.synchronize()
        assert torch.equal(ya.view(torch.int32), yb.view(torch.int32)), (seed, f0)
        got[f0:f0 + B] = ya.cpu().numpy().reshape(B, N)
    ref = graph_eval.run_graph(a.g, x[:, :8])
    if seed % 2:
        assert np.abs(got[:, :8] - ref).max() <= 1e-5 * np.abs(ref).max()
    else:
        assert ulp_diff(got[:, :8], ref).max() <= 1, seed
    a.close()
    b.close()


def test_without_the_kernel_headers_everything_still_runs(dspfx, G, monkeypatch, tmp_path):
    """The run-time compiler needs the kernel headers next to the library (or in DSPFX_KERNEL_HEADERS).  Without them
    dspfx_graph_set reports DSPFX_ERR_UNSUPPORTED and the graph is evaluated run by run on the interpreter kernels --
    same bits as with them."""
    import torch
    E = dspfx
    N, B = 256, 128
    good = G.GraphEngine(graphs.fan_in_three(), N, B)
    assert good.fused is not None
    monkeypatch.setenv("DSPFX_KERNEL_HEADERS", str(tmp_path))           # an empty directory
    monkeypatch.setenv("DSPFX_JIT", "1")
    eng = E.Engine(N, B)
    specs, links = G.fused_plan(G.Graph(graphs.random_dag(4242, 7)))
    with pytest.raises(E.DspfxError) as ei:
        eng.set_graph(specs, links)
    assert ei.value.status == E.ERR_UNSUPPORTED and "compiled" in str(ei.value)
    eng.set_chain([E.Gain(0.5), E.LowPass(0.25), E.HighPass(0.125)])   # a chain shape nothing else uses: not in the cache either
    assert "jit_" not in eng.describe() and "dyn" in eng.describe(), eng.describe()
    eng.close()
    bad = G.GraphEngine(graphs.random_dag(4243, 7), N, B)               # falls back by itself
    assert bad.fused is None and bad.runs
    monkeypatch.delenv("DSPFX_KERNEL_HEADERS")
    ref = G.GraphEngine(graphs.random_dag(4243, 7), N, B, fused=True)
    x = torch.empty(B * N, dtype=torch.float32, device="cuda")
    for k in range(3):
        ref.util.fill_noise(x, B, k * B, 0x5EED0010)
        ya = bad.process(x, B).clone()
        yb = ref.process(x, B)
        torch.cuda.synchronize()
        assert torch.equal(ya.view(torch.int32), yb.view(torch.int32)), k
    for e in (good, bad, ref):
        e.close()


def test_unplugged_ports_give_positive_zero(dspfx):
    """An effect with nothing plugged in feeds the Output node: the reference's arithmetic gives +0.0 (0 - (+0) = +0, then
    0 + that, then the division).  With a literal zero in the generated code the compiler turned `(0 - z) / c` into
    `(-z) / c` = -0.0 (found by tools/graph_sweep.py, seed 2044); unconnected ports are opaque zeros since."""
    import torch
    E = dspfx
    N, B = 64, 128
    x = torch.zeros(B * N, dtype=torch.float32, device="cuda")
    y = torch.empty_like(x)
    hp, sq, si = E.HighPass(0.5), E.SignalGen(-0.5, 9000.0, E.SIG_SQUARE), E.SignalGen(0.7, 220.0, E.SIG_SINE)
    M, S = E.PORT_MAIN, E.PORT_SLIDER
    cases = [([hp], [(0, 1, M)]),
             ([hp, sq, si], [(0, 1, S), (1, 2, S), (0, 3, M), (2, 3, M)]),                 # generators whose amplitude is driven to 0: +-0 terms
             ([hp, sq, si], [(0, 1, S), (1, 2, S), (0, 3, M), (1, 3, M), (2, 3, M)]),
             ([E.Gain(2.0), hp], [(E.GRAPH_ZERO, 0, M), (0, 1, M), (1, 2, M)])]           # a connected pipe of zeros (demux) in front
    for specs, links in cases:
        eng = E.Engine(N, B)
        eng.set_graph(specs, links)
        for k in range(3):
            eng.process(x, out=y, n_frames=B)
            torch.cuda.synchronize()
            assert int((y.view(torch.int32) != 0).sum()) == 0, (len(specs), k)     # every sample is +0.0: no sign bit, no value
        eng.close()


def test_graph_set_rejections(dspfx):
    E = dspfx
    eng = E.Engine(128, 128)
    gain = E.NodeSpec(E.GAIN, [0.5])
    with pytest.raises(E.DspfxError) as ei:      # a link must go forward
        eng.set_graph([gain, gain], [(1, 0, E.PORT_MAIN)])
    assert ei.value.status == -1
    with pytest.raises(E.DspfxError) as ei:      # GAIN has one slider port and no "b" port
        eng.set_graph([gain], [(E.GRAPH_INPUT, 0, E.PORT_SIDE)])
    assert ei.value.status == -1
    with pytest.raises(E.DspfxError) as ei:
        eng.set_graph([gain], [(E.GRAPH_INPUT, 0, E.PORT_SLIDER + 1)])
    assert ei.value.status == -1
    with pytest.raises(E.DspfxError) as ei:      # a RAW link is its port's only link
        eng.set_graph([gain, gain], [(E.GRAPH_INPUT, 0, E.PORT_MAIN), (0, 1, E.PORT_MAIN | E.PORT_RAW), (E.GRAPH_INPUT, 1, E.PORT_MAIN)])
    assert ei.value.status == -1
    with pytest.raises(E.DspfxError) as ei:      # Fuzz has its own kernel
        eng.set_graph([E.NodeSpec(E.DISTORT, [2.0], mode=E.FUZZ)], [(E.GRAPH_INPUT, 0, E.PORT_MAIN)])
    assert ei.value.status == E.ERR_UNSUPPORTED
    with pytest.raises(E.DspfxError) as ei:
        eng.set_graph([gain] * 17, [])
    assert ei.value.status == E.ERR_UNSUPPORTED
    eng.close()
    odd = E.Engine(100, 128)
    with pytest.raises(E.DspfxError) as ei:      # whole waves only
        odd.set_graph([gain], [(E.GRAPH_INPUT, 0, E.PORT_MAIN), (0, 1, E.PORT_MAIN)])
    assert ei.value.status == E.ERR_UNSUPPORTED
    odd.set_chain([gain])                        # the engine is still a usable chain engine
    odd.close()


def test_graph_partition_and_rejections(dspfx, G):
    g = G.GraphEngine(graphs.diamond(), 64, fused=False)
    runs = sorted(sorted(m.id for m in r.nodes) for r in g.runs)
    # gain (fan-out 2) and high_pass (fan-out 2) end their runs; biquad > high_pass fuse; add starts a run
    # (two different producers), distort has fan-in, distort > reverb fuse
    assert runs == [[1], [2, 3], [4], [5, 6]], runs
    g.close()
    import json
    cyc = json.loads(graphs.diamond())
    # reverb(6) -> gain(1).in closes a cycle 1 -> 4 -> 5 -> 6 -> 1
    n6 = next(n for n in cyc["nodes"] if n["id"] == 6)
    n1 = next(n for n in cyc["nodes"] if n["id"] == 1)
    cyc["links"].append({"lhs": [6, n6["cfg"]["outputs"]["out"]], "rhs": [1, n1["cfg"]["inputs"]["in"]]})
    with pytest.raises(Exception) as ei:
        G.Graph(json.dumps(cyc))
    assert "cycle" in str(ei.value)
    two = json.loads(graphs.lfo_tremolo())
    n0 = next(n for n in two["nodes"] if n["id"] == 0)
    n2 = next(n for n in two["nodes"] if n["id"] == 2)
    two["links"].append({"lhs": [0, n0["cfg"]["outputs"]["out"]], "rhs": [2, n2["cfg"]["inputs"]["level"]]})
    with pytest.raises(Exception) as ei:
        G.Graph(json.dumps(two))
    assert "averages several links" in str(ei.value)


@pytest.mark.parametrize("seed,n", [(1, 40), (2, 40), (13, 23)])
def test_region_plan_equals_run_by_run_and_the_oracle(dspfx, G, seed, n):
    """A 40-node random DAG has no single-signal cut: it runs as three generated kernels exchanging several blocks
    (region_plan + dspfx_process_io).  Same f32 operations in the same order as the run-by-run evaluation: bit-identical;
    <= 1 ulp from the oracle (exact-arithmetic kinds)."""
    import torch
    N, B, blocks = 4096, 128, 4
    text = graphs.random_dag(seed, n)
    a = G.GraphEngine(text, N, B, regions=True)
    b = G.GraphEngine(text, N, B, fused=False)
    assert a.regions and not b.regions and sum(1 for st in a.regions if st["kind"] == "region") <= -(-n // 16)
    x = O.noise(0x5EED0040 + seed, np.arange(N), np.arange(B * blocks))
    got = np.empty_like(x)
    for k in range(blocks):
        dx = torch.from_numpy(x[k * B:(k + 1) * B]).cuda()
        ya, yb = a.process(dx, B), b.process(dx, B)
        torch.cuda.synchronize()
        assert torch.equal(ya, yb), (k, a.describe())
        got[k * B:(k + 1) * B] = ya.cpu().numpy().reshape(B, N)
    chans = np.r_[0:24, N - 8:N]
    ref = graph_eval.run_graph(a.g, x[:, chans])
    assert ulp_diff(got[:, chans], ref).max() <= 1
    # the default selection picks the region plan for this graph too (no one-kernel, no series form)
    c = G.GraphEngine(text, N, B)
    assert c.regions and c.fused is None and not c.series
    for e in (a, b, c):
        e.close()


def test_signals_around_fir_and_fuzz_nodes_with_a_node_fed_level_port(dspfx, G):
    """Several signals bypass a FIR node and a Fuzz node whose level slider is driven by an LFO: no series form; the
    region plan keeps both nodes as steps of their own between generated kernels.  Equal to run by run within the Fuzz /
    FIR bars, and to the oracle."""
    import torch
    N, B, blocks = 2048, 128, 5
    text = graphs.around_fir_and_fuzz()
    a = G.GraphEngine(text, N, B)
    b = G.GraphEngine(text, N, B, fused=False)
    assert a.regions and [st["kind"] for st in a.regions].count("node") == 2, a.describe()
    x = O.noise(0x5EED0051, np.arange(N), np.arange(B * blocks))
    ga, gb = np.empty_like(x), np.empty_like(x)
    for k in range(blocks):
        dx = torch.from_numpy(x[k * B:(k + 1) * B]).cuda()
        ya, yb = a.process(dx, B), b.process(dx, B)
        torch.cuda.synchronize()
        ga[k * B:(k + 1) * B] = ya.cpu().numpy().reshape(B, N)
        gb[k * B:(k + 1) * B] = yb.cpu().numpy().reshape(B, N)
    assert np.array_equal(ga.view(np.uint32), gb.view(np.uint32))          # the same kernels for FIR / Fuzz, the same f32 ops around
    chans = np.r_[0:16, N - 4:N]
    ref = graph_eval.run_graph(a.g, x[:, chans])
    assert np.abs(ga[:, chans] - ref).max() <= 1e-5 * np.abs(ref).max()
    a.close()
    b.close()


def test_golden_graphs_through_the_region_plan(dspfx, G):
    """Every committed graph golden through the general region plan (forced): the oracle's outputs within the bars of
    test_graph_golden_vectors."""
    import torch
    from test_graph_cpu import _graph_goldens
    for name, doc, x, y in _graph_goldens():
        N = 64
        xx = np.tile(x, (1, -(-N // x.shape[1])))[:, :N].astype(F)
        ge = G.GraphEngine(doc, N, 128, regions=True)
        got = np.empty_like(xx)
        for f0 in range(0, xx.shape[0], 128):
            dx = torch.from_numpy(np.ascontiguousarray(xx[f0:f0 + 128])).cuda()
            out = ge.process(dx, 128)
            torch.cuda.synchronize()
            got[f0:f0 + 128] = out.cpu().numpy().reshape(128, N)
        ref = np.tile(y, (1, -(-N // y.shape[1])))[:, :N]
        assert np.abs(got - ref).max() <= 2e-5 * max(1.0, np.abs(ref).max()), (name, ge.describe())
        ge.close()


def test_the_pin_kits_cases_through_the_hip_path(dspfx, G):
    """oracle/pin_kit/cases.json is the set of cases the reference-side harness (golden_dump.rs) runs through the REAL nodes.  The same
    documents and the same samples through the HIP engine, against the same expected vectors with the same bars
    (tools/compare_pin.py's judge): when a maintainer's one command says the reference matches the vectors, this test has already
    said the GPU does -- reference == vectors == GPU on one case set.  (The two menu-fresh cases are chains:
    test_gpu_parity.py::test_reverb_fresh_from_the_menu_*; a document cannot say "fresh".)"""
    import json
    import os
    import sys
    import torch
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    for p in (os.path.join(root, "oracle", "pin_kit"), os.path.join(root, "tools")):
        if p not in sys.path:
            sys.path.insert(0, p)
    import compare_pin
    import export_cases
    N, B = 64, 128
    ran, forms = 0, {}
    for c in export_cases.build_cases():
        if c["fresh"]:
            continue
        ge = G.GraphEngine(json.dumps(c["doc"]), N, B)
        kind = "one kernel" if ge.fused is not None else ("series" if ge.series else ("regions" if ge.regions else "run by run"))
        forms[kind] = forms.get(kind, 0) + 1
        for ch, (xb, wb) in enumerate(zip(c["x"], c["want"])):
            if ch:
                ge.close()
                ge = G.GraphEngine(json.dumps(c["doc"]), N, B)          # every channel of a case is an independent run
            x = compare_pin.f32(xb)
            want = compare_pin.f32(wb)
            got = np.empty_like(x)
            for f0 in range(0, x.size, B):
                dx = torch.from_numpy(np.repeat(x[f0:f0 + B, None], N, axis=1).copy()).cuda()
                y = ge.process(dx, B)
                torch.cuda.synchronize()
                yy = y.cpu().numpy().reshape(B, N)
                assert np.array_equal(yy.view(np.uint32), np.repeat(yy[:, :1], N, axis=1).view(np.uint32)), c["name"]     # identical channels
                got[f0:f0 + B] = yy[:, 0]
            bar = c["bar"]
            if bar["kind"] == "ulp" and any(n["typename"] == "fir" for n in c["doc"]["nodes"]) and c["name"] != "chain_fir_int5":
                bar = dict(kind="rel_peak", tol=1e-5)       # the MFMA sweeps' stated bar (test_graph_golden_vectors); the reference's f64 loop is held to ulps
            ok, text = compare_pin.judge(got, want, bar)
            assert ok, (c["name"], ch, text, kind)
        ge.close()
        ran += 1
    assert ran >= 54
    print("pin-kit cases through the HIP path: %d cases, %s" % (ran, forms))
