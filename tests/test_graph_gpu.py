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


@pytest.mark.parametrize("name", ["diamond", "lfo_tremolo", "fan_in_three", "routing", "routing_ab", "routing_ba"])
@pytest.mark.parametrize("N,tile,B", [(100, 0, 128), (128, 64, 128), (64, 0, 256)])
def test_graph_matches_reference_semantics(dspfx, G, name, N, tile, B):
    import torch
    text = {"routing_ab": lambda: graphs.routing("A", "B"), "routing_ba": lambda: graphs.routing("B", "A")}.get(
        name, getattr(graphs, name, None))()
    nf = 768
    x = O.noise(0x5EED0001, np.arange(N), np.arange(nf))
    ge = G.GraphEngine(text, N, B, tile_channels=tile)
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


def test_graph_partition_and_rejections(dspfx, G):
    g = G.GraphEngine(graphs.diamond(), 64)
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
