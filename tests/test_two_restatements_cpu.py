"""The two CPU restatements of the reference -- the C oracle (oracle/dspfx_oracle.c) and the numpy model
(oracle/numpy_model.py), written separately from the cited Rust lines -- pushed through the same chains, control
ports, tap reloads and whole saved graphs.  The reference ships no vectors (parity unpinned upstream), so the only
pinning left is that two independent readings of the same lines agree; any disagreement here is an oracle bug.

Arithmetic nodes: bit for bit.  libm-backed nodes (tanh / sin / atan / exp: numpy's float32 routines vs glibc's) are
left out of the bit-exact sweeps and compared with a tolerance where they appear."""
import numpy as np

import graph_eval
import graphs
import numpy_model as M
import oracle as O

F = np.float32


def _noise(seed, n, ch=0):
    return O.noise(seed, np.array([ch]), np.arange(n))[:, 0]


def test_chain_runner_with_control_ports_and_latch():
    """lib.rs:122-161 through both: per-sample slider map, latch of each block's first value, the latch applying
    once the port is disconnected, a slider store overwriting it; every link flag setting; side input."""
    nf = 128 * 6
    x, side = _noise(1, nf), _noise(2, nf)
    sigs = {k: (_noise(10 + i, nf) * F(1.5)).astype(F) for i, k in enumerate([(0, 0), (1, 0), (2, 0), (4, 0), (5, 0), (5, 1)])}
    descs = [dict(kind=O.GAIN, params=[1.0]), dict(kind=O.DISTORT, params=[3.0], mode=O.HARD_CLIP),
             dict(kind=O.MIX, params=[0.5]), dict(kind=O.BIQUAD, params=[1.0, -1.2, 0.5, 0.3, 0.2, 0.1]),
             dict(kind=O.DISTORT, params=[2.0], mode=O.SOFT_CLIP), dict(kind=O.SIGNAL_GEN, params=[0.4, 300.0], mode=O.SIG_TRIANGLE),
             dict(kind=O.ADD), dict(kind=O.ENVELOPE, params=[3.0, 80.0])]
    for lf in (0, 1, 2, 3):
        cn = [O.node_from_desc(d) for d in descs]
        mn = [M.make_node(d) for d in descs]
        keys = list(sigs)
        plan = [keys, keys, keys[:2], [], [], keys[-2:]]
        for b, ks in enumerate(plan):
            sl = slice(128 * b, 128 * (b + 1))
            if b == 4:                                   # a slider store overwrites the latch (lib.rs:487-492)
                cn[0].set_param(0, 0.7)
                mn[0].set_param(0, 0.7)
            ctl = {k: sigs[k][sl] for k in ks} or None
            yc = O.chain_run(cn, x[sl], lf, side[sl], ctl=ctl)
            ym = M.chain_run(mn, x[sl], lf, side[sl], ctl=ctl)
            assert np.array_equal(yc.view(np.uint32), ym.view(np.uint32)), (lf, b)


def test_fuzz_with_level_port_in_both():
    """distort.rs:146-172,176-180: the level port is mapped for Fuzz too and zipped per sample.  exp differs between
    glibc and numpy by an ulp, amplified by the three block-global normalisations: the Fuzz bar of the GPU tests."""
    nf = 256
    x = _noise(3, nf)
    sig = (_noise(4, nf) * F(1.4)).astype(F)
    d = [dict(kind=O.DISTORT, params=[3.0], mode=O.FUZZ)]
    for ctl in (None, {(0, 0): sig}):
        yc = O.chain_run([O.node_from_desc(d[0])], x, 3, ctl=ctl)
        ym = M.chain_run([M.make_node(d[0])], x, 3, ctl=ctl)
        assert np.abs(yc - ym).max() <= 4e-6 * np.abs(yc).max()
    assert np.abs(O.chain_run([O.node_from_desc(d[0])], x, 3) - O.chain_run([O.node_from_desc(d[0])], x, 3, ctl={(0, 0): sig})).max() > 1e-3


def test_fir_tap_reload_keeps_the_history():
    """fir.rs:153-171 replaces `taps` only; `state` (64-65) is never cleared and at most one sample is popped per step
    (193-197).  Derived from those lines alone: after a reload with FEWER taps the deque stays at its old length L, so
    the taps pair with its OLDEST samples: y[n] = sum_k taps_rev[k] x[n-L+1+k] -- the plain convolution delayed by
    L - T samples; with MORE taps the deque goes on growing front-aligned.  Both restatements, bit for bit."""
    rng = np.random.default_rng(5)
    x = rng.integers(-8, 8, 400).astype(F)                # integers: every sum is exact, no rounding to argue about
    h1, h2 = np.array([1, 2, 3, 4, 5, 6, 7, 8], np.float64), np.array([3, -1, 2], np.float64)
    for cls in ("c", "numpy"):
        n = O.Node(O.FIR, taps_reversed=h1[::-1]) if cls == "c" else M.Fir(h1[::-1])
        run = (lambda v: np.concatenate([n.process(v[i:i + 128]) for i in range(0, len(v), 128)])) if cls == "c" else n.run
        y1 = run(x[:200])
        assert np.array_equal(y1[7:], np.convolve(x[:200].astype(np.float64), h1)[7:200].astype(F))
        n.set_taps(h2[::-1])                               # L = 8 samples held, T' = 3
        y2 = run(x[200:])
        full = np.convolve(x.astype(np.float64), h2)[:400]
        delay = 8 - 3
        assert np.array_equal(y2, full[200 - delay:400 - delay].astype(F)), cls
        # and growing again: 3 -> 6 taps with 8 samples held: still longer than T', still a pure delay (of 2)
        h3 = np.array([1, 0, -2, 4, 1, 1], np.float64)
        n.set_taps(h3[::-1])
        xx = rng.integers(-8, 8, 100).astype(F)
        y3 = run(xx)
        allx = np.concatenate([x, xx]).astype(np.float64)
        assert np.array_equal(y3, np.convolve(allx, h3)[400 - 2:500 - 2].astype(F)), cls
    # random taps, every growth / shrink pattern, the a/b slice split included: the two restatements agree bit for bit
    for T1, T2, T3 in ((8, 3, 40), (3, 70, 5), (33, 33, 2), (5, 6, 7), (64, 9, 130)):
        hs = [rng.uniform(-1, 1, T) for T in (T1, T2, T3)]
        xs = rng.uniform(-1, 1, 3 * 150).astype(F)
        c, m = O.Node(O.FIR, taps_reversed=hs[0][::-1]), M.Fir(hs[0][::-1])
        for k in range(3):
            if k:
                c.set_taps(hs[k][::-1])
                m.set_taps(hs[k][::-1])
            seg = xs[150 * k:150 * (k + 1)]
            yc = np.concatenate([c.process(seg[:128]), c.process(seg[128:])])
            assert np.array_equal(yc.view(np.uint32), m.run(seg).view(np.uint32)), (T1, T2, T3, k)


def _graph(text):
    from __graft_entry__ import load_package
    load_package()
    from dsp_stuff_amd import graph as G
    return G.Graph(text)


def test_golden_graphs_through_both_restatements():
    """Every committed graph golden (tests/golden/graphs): the numpy evaluator (M.run_graph: its own node models, its
    own collect_and_average) against the C-oracle evaluator and the committed outputs.  Graphs without libm nodes
    must agree bit for bit; FIR graphs too (f64 sums in the same order)."""
    from test_graph_cpu import _graph_goldens
    seen = 0
    for name, doc, x, y in _graph_goldens():
        g = _graph(doc)
        libm = any(n.spec is not None and (n.spec.kind in (O.OVERDRIVE, O.CHEBYSHEV) or
                                          (n.spec.kind == O.DISTORT and n.spec.mode in (O.TANH, O.SIN, O.ATAN, O.FUZZ)) or
                                          (n.spec.kind == O.SIGNAL_GEN and n.spec.mode == O.SIG_SINE) or
                                          (n.spec.kind == O.ENVELOPE)) for n in g.nodes.values())
        xs = x[:256, :2]
        ym = M.run_graph(g, xs)
        yc = graph_eval.run_graph(g, xs)
        assert np.array_equal(yc, y[:256, :2]), name
        if libm:
            assert np.abs(ym - yc).max() <= 3e-6 * max(1.0, np.abs(yc).max()), name
        else:
            assert np.array_equal(ym.view(np.uint32), yc.view(np.uint32)), name
        seen += 1
    assert seen >= 8


def test_random_dags_through_both_restatements():
    """Seeded random DAGs (fan-in, fan-out, unplugged ports, slider ports fed by nodes, generators): exact-arithmetic
    kinds only, so the two evaluators must agree bit for bit -- DAG order, `as_input` latching and the averaging of
    every port rest on two restatements now, not one."""
    for seed in range(24):
        n_nodes = 4 + seed % 9
        g = _graph(graphs.random_dag(1000 + seed, n_nodes))
        x = O.noise(77 + seed, np.arange(2), np.arange(384))
        yc = graph_eval.run_graph(g, x)
        ym = M.run_graph(g, x)
        has_env = any(n.spec is not None and n.spec.kind == O.ENVELOPE and (n.spec.params[0] or n.spec.params[1]) for n in g.nodes.values())
        if has_env:   # envelope gains: glibc powf vs numpy's float64 power rounded once (<= 1 ulp apart)
            assert np.abs(yc - ym).max() <= 2e-6 * max(1.0, np.abs(yc).max()), seed
        else:
            assert np.array_equal(yc.view(np.uint32), ym.view(np.uint32)), (seed, n_nodes, np.abs(yc - ym).max())
