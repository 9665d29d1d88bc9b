"""Hand-built DSPConfig documents (the reference's saved-graph JSON, runtime.rs:606-612) for the DAG tests."""
import json

_PORTS = {   # typename -> (input port names in field order, has output)
    "input": ([], True), "output": (["in"], False),
    "gain": (["in", "level"], True), "biquad": (["in"], True), "low_pass": (["in"], True), "high_pass": (["in"], True),
    "reverb": (["in"], True), "distort": (["in", "level"], True), "overdrive": (["in", "boost", "drive", "level"], True),
    "chebyshev": (["in"], True), "add": (["a", "b"], True), "mix": (["a", "b", "ratio"], True),
    "signal_gen": (["amplitude", "frequency"], True), "envelope": (["in"], True),
    "mux": (["a", "b"], True), "demux": (["in"], True), "fir": (["in"], True),
}
_OUTS = {"demux": ["a", "b"]}


def build(nodes, links):
    """nodes: [(id, typename, {saved fields})]; links: [(src_id, dst_id, dst_port)] in the order the reference
    would hold them.  Port ids are assigned here."""
    next_id = [1000]

    def pid():
        next_id[0] += 1
        return next_id[0]

    doc_nodes, ports = [], {}
    for nid, tn, fields in nodes:
        ins, has_out = _PORTS[tn]
        cfg = {"id": nid, "inputs": {p: pid() for p in ins},
               "outputs": {o: pid() for o in _OUTS.get(tn, ["out"])} if has_out else {}}
        cfg.update(fields)
        ports[nid] = cfg
        doc_nodes.append({"id": nid, "typename": tn, "position": [0.0, 0.0], "cfg": cfg})
    doc_links = []
    for l in links:
        s, d, p = l[:3]
        o = l[3] if len(l) > 3 else "out"          # 4th element: the producer's output port (demux: "a" / "b")
        doc_links.append({"lhs": [s, ports[s]["outputs"][o]], "rhs": [d, ports[d]["inputs"][p]]})
    return json.dumps({"nodes": doc_nodes, "links": doc_links})


BQ = {"a0": 1.0, "a1": -1.2, "a2": 0.5, "b0": 0.3, "b1": 0.2, "b2": 0.1}


def diamond():
    """input -> gain -> (add.a);  input -> biquad -> lowpass -> (add.b);  add -> distort <- gain (2 links);
    distort and the biquad branch both into the output node."""
    return build(
        [(0, "input", {}), (1, "gain", {"level": 0.8}), (2, "biquad", BQ), (3, "high_pass", {"ratio": 0.3}),
         (4, "add", {}), (5, "distort", {"level": 3.0, "mode": "SoftClip"}), (6, "reverb", {"seconds": 0.004, "decay": 0.4}),
         (9, "output", {})],
        [(0, 1, "in"), (0, 2, "in"), (2, 3, "in"), (1, 4, "a"), (3, 4, "b"), (4, 5, "in"), (1, 5, "in"), (5, 6, "in"),
         (6, 9, "in"), (3, 9, "in")])


def lfo_tremolo():
    """A triangle LFO patched into a gain's level port; the envelope of the input drives a mix ratio."""
    return build(
        [(0, "input", {}), (1, "signal_gen", {"amplitude": 0.9, "frequency": 600.0, "mode": "Triangle"}),
         (2, "gain", {"level": 1.0}), (3, "envelope", {"attack": 4.0, "release": 200.0}), (4, "biquad", BQ),
         (5, "mix", {"ratio": 0.5}), (9, "output", {})],
        [(0, 2, "in"), (1, 2, "level"), (0, 3, "in"), (0, 4, "in"), (2, 5, "a"), (4, 5, "b"), (3, 5, "ratio"), (5, 9, "in")])


def fan_in_three():
    """Three branches averaged on one port (k = 3), a mix whose "b" port averages two links, an effect with
    nothing plugged in, and a generator-only branch."""
    return build(
        [(0, "input", {}), (1, "gain", {"level": 0.5}), (2, "low_pass", {"ratio": 0.6}), (3, "high_pass", {"ratio": 0.2}),
         (4, "distort", {"level": 2.0, "mode": "HardClip"}), (5, "mix", {"ratio": 0.25}),
         (6, "signal_gen", {"amplitude": 0.3, "frequency": 12000.0, "mode": "Square"}), (7, "gain", {"level": 2.0}),
         (8, "biquad", BQ), (9, "output", {})],
        [(0, 1, "in"), (0, 2, "in"), (0, 3, "in"), (1, 4, "in"), (2, 4, "in"), (3, 4, "in"), (4, 5, "a"), (2, 5, "b"),
         (6, 5, "b"), (6, 7, "in"), (5, 9, "in"), (7, 9, "in"), (8, 9, "in")])


def routing(in_port="B", out_port="A"):
    """mux picks one of two branches; demux sends the signal to one of two effects, both averaged by the output
    (the unselected one delivers zeros but still counts as a connected pipe)."""
    return build(
        [(0, "input", {}), (1, "gain", {"level": 0.5}), (2, "low_pass", {"ratio": 0.4}), (3, "mux", {"in_port": in_port}),
         (4, "demux", {"out_port": out_port}), (5, "distort", {"level": 2.0, "mode": "HardClip"}), (6, "biquad", BQ),
         (9, "output", {})],
        [(0, 1, "in"), (0, 2, "in"), (1, 3, "a"), (2, 3, "b"), (3, 4, "in"), (4, 5, "in", "a"), (4, 6, "in", "b"),
         (5, 9, "in"), (6, 9, "in")])


def random_dag(seed, n_nodes=8, libm=False):
    """A random DAG of `n_nodes` effect nodes between an input (id 0) and an output (id 99): every port draws 0-3
    links from earlier nodes (fan-in, fan-out and unplugged ports all occur), slider ports at most one.
    libm=False keeps to the kinds whose arithmetic is exact f32 (no tanh / sin / exp), so the oracle comparison
    needs no tolerance beyond the hop's."""
    import random
    rnd = random.Random(seed)
    exact_modes = ["HardClip", "SoftClip", "RecipSoftClip", "Square", "Chebyshev4"]
    pool = ["gain", "biquad", "low_pass", "high_pass", "reverb", "distort", "add", "mix", "envelope", "signal_gen", "gain", "mix"]
    if libm:
        pool += ["overdrive", "chebyshev", "distort_libm", "signal_sine"]
    nodes, links = [(0, "input", {})], []
    for nid in range(1, n_nodes + 1):
        tn = rnd.choice(pool)
        if tn == "gain":
            f = {"level": rnd.choice([0.5, 0.8, 1.0, 1.25])}
        elif tn == "biquad":
            f = dict(BQ)
        elif tn in ("low_pass", "high_pass"):
            f = {"ratio": rnd.choice([0.2, 0.5, 0.7])}
        elif tn == "reverb":
            f = {"seconds": rnd.choice([0.003, 0.004, 0.01]), "decay": rnd.choice([0.3, 0.5])}
        elif tn == "distort":
            f = {"level": rnd.choice([0.0, 1.5, 3.0, 7.0]), "mode": rnd.choice(exact_modes)}
        elif tn == "distort_libm":
            tn, f = "distort", {"level": rnd.choice([1.5, 3.0]), "mode": rnd.choice(["Tanh", "Sin", "Atan"])}
        elif tn == "mix":
            f = {"ratio": rnd.choice([0.0, 0.25, 0.5, 1.0])}
        elif tn == "envelope":
            f = {"attack": rnd.choice([0.0, 4.0, 50.0]), "release": rnd.choice([0.0, 100.0, 400.0])}
        elif tn == "signal_gen":
            f = {"amplitude": rnd.choice([0.3, 0.9, -0.5]), "frequency": rnd.choice([50.0, 440.0, 9000.0]),
                 "mode": rnd.choice(["Triangle", "Square", "Constant"])}
        elif tn == "signal_sine":
            tn, f = "signal_gen", {"amplitude": 0.7, "frequency": rnd.choice([220.0, 3000.0]), "mode": "Sine"}
        elif tn == "overdrive":
            f = {"boost": rnd.choice([2.0, 10.0]), "drive": rnd.choice([0.3, 0.8]), "level": rnd.choice([0.0, 0.5, 1.0])}
        elif tn == "chebyshev":
            f = {"level_pos": rnd.choice([0.5, 2.0]), "level_neg": rnd.choice([0.7, 3.0])}
        else:
            f = {}
        nodes.append((nid, tn, f))
        ins = _PORTS.get(tn, (["in"], True))[0]
        earlier = list(range(0, nid))
        for k, port in enumerate(ins):
            is_signal_port = port in ("in", "a", "b")
            if is_signal_port:
                n_links = rnd.choice([0, 1, 1, 1, 2, 3]) if tn != "signal_gen" else 0
            else:
                n_links = rnd.choice([0, 0, 1])
            for s in rnd.sample(earlier, min(n_links, len(earlier))):
                links.append((s, nid, port))
    nodes.append((99, "output", {}))
    for s in rnd.sample(range(1, n_nodes + 1), rnd.choice([1, 2, 3])):
        links.append((s, 99, "in"))
    if not any(l[1] == 99 and l[0] == n_nodes for l in links):
        links.append((n_nodes, 99, "in"))       # the last node always reaches the output
    return build(nodes, links)


def cab_rig(taps=64, bypass=False, cut="fir", dry=False):
    """The shape of a guitar rig: drive stage with a parallel clean path -> cabinet impulse response (FIR) -> Fuzz-free
    tail with a delay mixed against the dry signal.  All the signal passes through the FIR node, so the graph is two
    fusable segments around it.  bypass=True adds a link around the FIR node (then it is not); cut="fuzz" puts a
    Distort/Fuzz node (block-global, also a kernel of its own) in the FIR node's place.  dry=True: the cabinet is fed by
    the Add alone and the same signal is mixed back in after it (wet / dry): one signal around the FIR node."""
    import math
    h = [math.exp(-6.0 * j / taps) * (1.0 if j % 3 else -0.7) / 4.0 for j in range(taps)]
    nodes = [(0, "input", {}), (1, "gain", {"level": 1.5}), (2, "distort", {"level": 4.0, "mode": "SoftClip"}),
             (3, "biquad", BQ), (4, "add", {}),
             (5, "fir", {"taps": h[::-1], "mode": "Balanced", "file_name": None}) if cut == "fir" else (5, "distort", {"level": 5.0, "mode": "Fuzz"}),
             (6, "reverb", {"seconds": 0.005, "decay": 0.4}), (7, "mix", {"ratio": 0.3}), (8, "high_pass", {"ratio": 0.1}),
             (9, "output", {})]
    links = [(0, 1, "in"), (1, 2, "in"), (1, 3, "in"), (2, 4, "a"), (3, 4, "b"), (4, 5, "in"), (2, 5, "in"),
             (5, 6, "in"), (5, 7, "a"), (6, 7, "b"), (7, 8, "in"), (8, 9, "in")]
    if bypass:
        links.append((4, 7, "a"))
    if dry:
        links = [l for l in links if l[:2] not in ((2, 5), (5, 7))] + [(6, 7, "a"), (4, 7, "b")]
        links = [l for l in links if l != (6, 7, "b")]
    return build(nodes, links)


def long_rig(seed, n_blocks=10, fir_at=None, dry_mix=False):
    """A pedalboard longer than one kernel holds: `n_blocks` stages in series, each a small sub-graph with one way in and
    one way out (a few effects in a row; two parallel branches into an Add or Mix; an effect whose slider is driven by an
    LFO; two branches averaged on one port).  One signal crosses between stages, so the graph can be cut there.
    fir_at=k puts a short FIR node after stage k; dry_mix=True ends in a Mix of the rig's output with the graph's Input
    (wet / dry over the whole rig: the Input stays alive beside every stage)."""
    import math
    import random
    rnd = random.Random(seed)
    nodes, links = [(0, "input", {})], []
    nid = [0]

    def add(tn, f):
        nid[0] += 1
        nodes.append((nid[0], tn, f))
        return nid[0]

    def effect():
        tn = rnd.choice(["gain", "biquad", "low_pass", "high_pass", "distort", "reverb", "envelope"])
        f = {"gain": {"level": rnd.choice([0.7, 1.0, 1.2])}, "biquad": dict(BQ), "low_pass": {"ratio": rnd.choice([0.3, 0.6])},
             "high_pass": {"ratio": rnd.choice([0.2, 0.5])},
             "distort": {"level": rnd.choice([1.5, 3.0]), "mode": rnd.choice(["HardClip", "SoftClip", "RecipSoftClip"])},
             "reverb": {"seconds": rnd.choice([0.003, 0.006]), "decay": rnd.choice([0.3, 0.5])},
             "envelope": {"attack": 4.0, "release": 100.0}}[tn]
        return add(tn, f)

    s = 0
    for b in range(n_blocks):
        kind = rnd.choice(["row", "row", "parallel", "lfo", "fan_in"])
        if kind == "row":
            for _ in range(rnd.choice([1, 2, 3])):
                e = effect()
                links.append((s, e, "in"))
                s = e
        elif kind == "parallel":
            a, c = effect(), effect()
            links += [(s, a, "in"), (s, c, "in")]
            m = add("mix", {"ratio": rnd.choice([0.25, 0.5])}) if rnd.random() < 0.5 else add("add", {})
            links += [(a, m, "a"), (c, m, "b")]
            s = m
        elif kind == "lfo":
            l = add("signal_gen", {"amplitude": 0.8, "frequency": rnd.choice([3.0, 440.0]), "mode": rnd.choice(["Triangle", "Square"])})
            gn = add("gain", {"level": 1.0})
            links += [(s, gn, "in"), (l, gn, "level")]
            s = gn
        else:
            a, c, d = effect(), effect(), effect()
            links += [(s, a, "in"), (s, c, "in"), (a, d, "in"), (c, d, "in")]
            s = d
        if fir_at == b:
            h = [math.exp(-5.0 * j / 32) * (1.0 if j % 2 else -0.5) / 3.0 for j in range(32)]
            f = add("fir", {"taps": h[::-1], "mode": "Balanced", "file_name": None})
            links.append((s, f, "in"))
            s = f
    if dry_mix:
        m = add("mix", {"ratio": 0.35})
        links += [(s, m, "a"), (0, m, "b")]
        s = m
    nodes.append((999, "output", {}))
    links.append((s, 999, "in"))
    return build(nodes, links)


def around_fir_and_fuzz(taps=48):
    """What the series form cannot express: several signals go AROUND a FIR node and a Fuzz node, and the Fuzz node's
    level slider is driven by an LFO (distort.rs:176-180 maps the level port for every mode).
    input -> gain -> fuzz(level <- lfo) -> fir -> add.a ; gain -> biquad -> add.b ; mix(a <- add, b <- fuzz) ; output <- mix, gain."""
    import math
    h = [math.exp(-0.08 * j) * (1 if j % 3 else -0.6) for j in range(taps)]
    return build(
        [(0, "input", {}), (1, "gain", {"level": 0.8}), (2, "signal_gen", {"amplitude": 0.6, "frequency": 300.0, "mode": "Triangle"}),
         (3, "distort", {"level": 3.0, "mode": "Fuzz"}), (4, "fir", {"mode": "Balanced", "file_name": None, "taps": h[::-1]}),
         (5, "biquad", BQ), (6, "add", {}), (7, "mix", {"ratio": 0.4}), (9, "output", {})],
        [(0, 1, "in"), (1, 3, "in"), (2, 3, "level"), (3, 4, "in"), (1, 5, "in"), (4, 6, "a"), (5, 6, "b"), (6, 7, "a"), (3, 7, "b"),
         (7, 9, "in"), (1, 9, "in")])
