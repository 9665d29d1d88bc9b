"""Parity at BASELINE.json's full sizes (configs 2-5), through checks whose cost does not grow
with the channel count: the synthetic input is an integer hash of (channel, sample index), so any
sampled channel can be regenerated and pushed through the oracle without shipping GiBs; plus
size-independent properties (checksum of the fused mix bus against an independent device-side sum,
exact linearity under power-of-two scaling, the delay line's impulse response)."""
import numpy as np
import pytest

import oracle as O
from chains import chain3, chain5, fir_taps, ulp_diff

pytestmark = pytest.mark.gpu
F = np.float32
SEED = 0x5EED0001


@pytest.fixture(scope="module")
def tc():
    import torch
    assert torch.cuda.is_available()
    return torch


def sample_channels(N, W=256):
    """First/last channel, wave/workgroup/tile edges, and a few in the middle."""
    c = {0, 1, 63, 64, 127, 128, W - 1, W, 2 * W - 1, N // 2 - 1, N // 2, N - W, N - 65, N - 64, N - 2, N - 1, 12345 % N, 777777 % N}
    return sorted(x for x in c if 0 <= x < N)


def oracle_channels(chain, chans, n_blocks, link_flags=3):
    descs = [n.oracle_desc() for n in chain]
    out = np.empty((n_blocks * 128, len(chans)), F)
    for i, c in enumerate(chans):
        y, _ = O.run_noise_channels(descs, SEED, c, 1, 0, n_blocks, link_flags=link_flags)
        out[:, i] = y[:, 0]
    return out


def gather(dspfx, t, chans, n_frames, N, tile):
    """Pull the sampled channels out of a device block in the engine layout -> [n_frames][len(chans)]."""
    import torch
    idx = torch.tensor(chans, device=t.device, dtype=torch.long)
    if tile:
        v = t.view(N // tile, n_frames, tile)
        return v[idx // tile, :, idx % tile].transpose(0, 1).contiguous().cpu().numpy()
    return t.view(n_frames, N)[:, idx].cpu().numpy()


def run_noise_engine(dspfx, tc, chain, N, B, n_calls, tile, chans, want_mix=False, info=None):
    eng = dspfx.Engine(N, B, link_flags=3, tile_channels=tile)
    eng.set_chain(chain)
    x = tc.empty(B * N, dtype=tc.float32, device="cuda")
    y = tc.empty_like(x)
    mix = tc.empty(B, dtype=tc.float32, device="cuda") if want_mix else None
    got = np.empty((n_calls * B, len(chans)), F)
    mix_err = 0.0
    for k in range(n_calls):
        eng.fill_noise(x, B, k * B, SEED)
        eng.process(x, out=y, mix=mix, n_frames=B)
        got[k * B:(k + 1) * B] = gather(dspfx, y, chans, B, N, tile)
        if want_mix and k % 16 == 0:      # checksum of the fused reduction vs an independent f64 sum on the device
            v = y.view(N // tile, B, tile) if tile else y.view(B, N)
            ref = v.double().sum(dim=(0, 2)) if tile else v.double().sum(dim=1)
            scale = float(ref.abs().max()) + 1.0
            mix_err = max(mix_err, float((mix.double() - ref).abs().max()) / scale)
    tc.cuda.synchronize()
    if info is not None:
        info["describe"] = eng.describe()
    eng.close()
    return got, mix_err


def test_config2_65536_channels_chain3(dspfx, tc):
    """BASELINE config 2: 65 536 channels, gain -> biquad -> delay(24000), 200 blocks (> one delay period)."""
    N, blocks = 1 << 16, 200
    chans = sample_channels(N)
    chain = chain3(dspfx)
    got, _ = run_noise_engine(dspfx, tc, chain, N, 128, blocks, 0, chans)
    ref = oracle_channels(chain, chans, blocks)
    assert ulp_diff(got, ref).max() <= 1
    assert np.abs(ref[24000:]).max() > 0      # the feedback path was really exercised


@pytest.mark.parametrize("B,tile", [(256, 256), (128, 0)])
def test_config3_and_5_million_channels_chain5(dspfx, tc, B, tile):
    """BASELINE config 3 (B=256) / config 5's per-GPU shard (B=128): 1 048 576 channels, 5-node chain,
    D=24000 (a 94 GiB ring), run past one delay period."""
    N = 1 << 20
    n_calls = 25600 // B
    chans = sample_channels(N)
    chain = chain5(dspfx)
    got, mix_err = run_noise_engine(dspfx, tc, chain, N, B, n_calls, tile, chans, want_mix=True)
    ref = oracle_channels(chain, chans, 25600 // 128)
    assert ulp_diff(got, ref).max() <= 1
    assert mix_err < 1e-5, mix_err


@pytest.mark.parametrize("sweep", ["half", "split", "f32"])
def test_config4_fir_262144_channels(dspfx, tc, monkeypatch, sweep):
    """BASELINE config 4: 262 144 channels x 4096-tap FIR on MFMA, through warm-up into steady state.  `half` is the sweep
    that ships (fir_half_kernel, two-part f16: the kernel bench.py's cfg4 line times), `split` the bf16 x 3 sweep (its second
    pass, and round 3's default), `f32` the f32 matrix-pipe sweep (cfg4_f32)."""
    N, T, blocks = 1 << 18, 4096, 40
    chans = sample_channels(N)[::2]
    chain = [dspfx.Fir(fir_taps(T))]
    monkeypatch.delenv("DSPFX_FIR_SPLIT", raising=False)
    monkeypatch.delenv("DSPFX_FIR_HALF", raising=False)
    if sweep == "f32":
        monkeypatch.setenv("DSPFX_FIR_SPLIT", "0")
    if sweep == "split":
        monkeypatch.setenv("DSPFX_FIR_HALF", "0")
    info = {}
    got, _ = run_noise_engine(dspfx, tc, chain, N, 128, blocks, 256, chans, info=info)
    assert {"half": "fir_halfp_kernel", "split": "fir_split_kernel", "f32": "fir_skew_kernel"}[sweep] in info["describe"], info["describe"]
    ref = oracle_channels(chain, chans, blocks)
    err = got.astype(np.float64) - ref.astype(np.float64)
    rms = np.sqrt(np.mean(err ** 2)) / np.sqrt(np.mean(ref.astype(np.float64) ** 2))
    assert rms < 1e-6, rms
    assert np.abs(err).max() < 3e-5


def test_linearity_and_channel_independence_full_size(dspfx, tc):
    """gain -> biquad -> delay is linear: scaling the input by 2^-3 scales the output exactly
    (no rounding changes under power-of-two scaling); and a channel's output does not depend on
    which channels surround it (permuting the input channels permutes the output)."""
    N, B, calls = 1 << 20, 128, 6
    chain = [dspfx.Gain(0.8), dspfx.BiQuad(), dspfx.Reverb(delay_samples=256, decay=0.5)]
    outs = []
    perm = tc.randperm(N, device="cuda", generator=tc.Generator(device="cuda").manual_seed(1))
    for mode in ("plain", "scaled", "permuted"):
        eng = dspfx.Engine(N, B, link_flags=3)
        eng.set_chain(chain)
        x = tc.empty((B, N), dtype=tc.float32, device="cuda")
        y = tc.empty_like(x)
        acc = []
        for k in range(calls):
            eng.fill_noise(x, B, k * B, SEED)
            if mode == "scaled":
                x.mul_(0.125)
            if mode == "permuted":
                x.copy_(x[:, perm])
            eng.process(x, out=y, n_frames=B)
            acc.append(y.clone())
        outs.append(tc.cat(acc))
        eng.close()
    plain, scaled, permuted = outs
    assert tc.equal(plain * 0.125, scaled)
    assert tc.equal(plain[:, perm], permuted)


def test_delay_impulse_full_size(dspfx, tc):
    """Every one of 1 048 576 delay lines of 24 000 samples answers an impulse with decay^k at k*D, exactly."""
    N, B, D = 1 << 20, 128, 24000
    eng = dspfx.Engine(N, B, link_flags=0, tile_channels=256)
    eng.set_chain([dspfx.Reverb(delay_samples=D, decay=0.5)])
    x = tc.zeros(B * N, dtype=tc.float32, device="cuda")
    y = tc.empty_like(x)
    n_calls = (2 * D) // B + 2
    for k in range(n_calls):
        x.zero_()
        if k == 0:
            x.view(N // 256, B, 256)[:, 0, :] = 1.0
        eng.process(x, out=y, n_frames=B)
        v = y.view(N // 256, B, 256)
        expect = tc.zeros(B, dtype=tc.float32, device="cuda")
        for j in (0, 1, 2):
            if k * B <= j * D < (k + 1) * B:
                expect[j * D - k * B] = 0.5 ** j
        assert tc.equal(v, expect[None, :, None].expand_as(v)), k
    eng.close()


@pytest.mark.parametrize("name", ["diamond", "lfo_tremolo", "fan_in_three"])
def test_whole_graph_kernel_million_channels(dspfx, tc, name):
    """A saved DAG at 1 048 576 tiled channels as ONE generated kernel: sampled channels against the node-by-node
    oracle evaluation (<= 1 ulp), and the whole block bit-identical to the run-by-run evaluation of the same graph."""
    import graph_eval
    import graphs
    from dsp_stuff_amd import graph as G
    N, B, tile, blocks = 1 << 20, 128, 256, 6
    chans = sample_channels(N, tile)
    text = getattr(graphs, name)()
    one = G.GraphEngine(text, N, B, tile_channels=tile, fused=True)
    runs = G.GraphEngine(text, N, B, tile_channels=tile, fused=False)
    x = tc.empty(B * N, dtype=tc.float32, device="cuda")
    got = np.empty((blocks * B, len(chans)), F)
    for k in range(blocks):
        one.util.fill_noise(x, B, k * B, SEED)
        ya = one.process(x, B)
        yb = runs.process(x, B)
        assert tc.equal(ya.view(tc.int32), yb.view(tc.int32)), (name, k)
        got[k * B:(k + 1) * B] = gather(dspfx, ya, chans, B, N, tile)
    xs = O.noise(SEED, np.array(chans), np.arange(blocks * B))
    ref = graph_eval.run_graph(one.g, xs)
    assert ulp_diff(got, ref).max() <= 1
    one.close()
    runs.close()


def test_config5_as_specified_eight_shards_against_the_monolith(dspfx, tc):
    """BASELINE config 5 as specified -- 8 388 608 channels sharded 8 x 1 048 576, 5-node chain, B = 128, the mix bus
    all-reduced -- on ONE GPU: the eight per-GPU engines (channel_offset = rank * 1 048 576) run one after the other, their
    un-normalised buses are summed in rank order (what the all-reduce computes) and divided by f32(0.0001 + 8 388 608)
    (dspfx_mix_allreduce's Output hop, through a one-rank communicator); next to them ONE engine holds all 8 388 608
    channels.  Every shard's output block equals the monolith's channel slice bit for bit (channels are independent and the
    noise is keyed by the global channel index), the sharded bus equals the monolith's bus within the bus' bar (the
    summation is associated differently), and sampled channels -- first / last of every shard among them -- equal the
    oracle to <= 1 ulp.  The delay is 256 samples here (at 24 000 the monolith's ring alone would be 805 GB); the 24 000-sample
    ring of one shard is what test_config3_and_5_million_channels_chain5 runs."""
    ranks, Nr, B, blocks, tile = 8, 1 << 20, 128, 4, 256
    N = ranks * Nr
    chain = chain5(dspfx, 256)
    mono = dspfx.Engine(N, B, link_flags=3, tile_channels=tile)
    mono.set_chain(chain)
    xm = tc.empty(B * N, dtype=tc.float32, device="cuda")
    ym = tc.empty_like(xm)
    bus_m = tc.empty((blocks, B), dtype=tc.float32, device="cuda")
    ys_m = []
    chans = sorted({0, 1, Nr - 1, Nr, 3 * Nr + 12345, 5 * Nr - 1, 5 * Nr, N - Nr, N - 2, N - 1})
    got = np.empty((blocks * B, len(chans)), F)
    for k in range(blocks):
        mono.fill_noise(xm, B, k * B, SEED)
        mono.process_bus(xm, ym, bus_m[k], B, n_connected=N)
        got[k * B:(k + 1) * B] = gather(dspfx, ym, chans, B, N, tile)
        ys_m.append(ym.view(N // tile, B, tile).clone())
    tc.cuda.synchronize()
    mono.close()
    del xm, ym
    comm = dspfx.Comm(0, 1, 0, dspfx.comm_unique_id())           # RCCL really runs (one rank: the sum over ranks is the identity)
    acc = tc.zeros((blocks, B), dtype=tc.float64, device="cuda")
    acc32 = tc.zeros((blocks, B), dtype=tc.float32, device="cuda")
    x = tc.empty(B * Nr, dtype=tc.float32, device="cuda")
    y = tc.empty_like(x)
    part = tc.empty(B, dtype=tc.float32, device="cuda")
    last = None
    for r in range(ranks):
        eng = dspfx.Engine(Nr, B, link_flags=3, tile_channels=tile, channel_offset=r * Nr)
        eng.set_chain(chain)
        for k in range(blocks):
            eng.fill_noise(x, B, k * B, SEED)
            eng.process_bus(x, y, part, B, n_connected=0)        # the rank-local, un-normalised bus
            tc.cuda.synchronize()
            shard = ys_m[k][r * (Nr // tile):(r + 1) * (Nr // tile)]
            assert tc.equal(y.view(Nr // tile, B, tile), shard), (r, k)
            acc32[k] += part                                      # f32 sums in rank order, as a ring all-reduce over 8 ranks adds them
            acc[k] += part.double()
        if last is not None:
            last.close()
        last = eng
    for k in range(blocks):
        last.mix_allreduce(comm, acc32[k], B, n_connected=N)      # the Output hop with the GLOBAL channel count
    tc.cuda.synchronize()
    div = float(dspfx.link_divisor(N))
    assert div == 8388608.0                                       # f32(0.0001 + 2^23) == 2^23: the increment is below half an ulp
    want = (acc / div).float()
    assert tc.allclose(acc32, want, rtol=1e-6, atol=1e-9)
    scale = float(bus_m.abs().max())
    assert float((acc32 - bus_m).abs().max()) <= 1e-5 * scale, (float((acc32 - bus_m).abs().max()), scale)
    last.close()
    comm.close()
    ref = np.empty_like(got)
    descs = [n.oracle_desc() for n in chain]
    for i, c in enumerate(chans):
        yc, _ = O.run_noise_channels(descs, SEED, c, 1, 0, blocks, link_flags=3)
        ref[:, i] = yc[:, 0]
    assert ulp_diff(got, ref).max() <= 1


def test_seconds_slider_dragged_on_config5_shard_with_blocks_in_flight(dspfx, tc):
    """Reverb::refresh_seconds (reverb.rs:55-71; run by the generated render() on every frame a drag moves the slider,
    dsp-stuff-derive/src/lib.rs:560-568) on BASELINE config 5's shard: 1 048 576 channels x 5 nodes, the seconds slider
    dragged 0.5 -> 0.25 -> 0.75 -> 0.5 -> 0.75 (24000 -> 12000 -> 36000 -> 24000 -> 36000 samples: a shrink, a growth BEYOND
    the ring's capacity -- 94 more groups of 512 MiB, allocated by the storing thread --, a shrink, a growth WITHIN capacity)
    from a second thread while 64 blocks are queued behind a busy stream.  Every store lands at a block boundary without the
    thread that drives the blocks waiting for the device (the stream is still busy when the call that applied it returns);
    sampled channels equal the oracle, with the stores at the blocks the engine logged, to <= 1 ulp."""
    import threading
    import time
    from test_gpu_threads import _hold
    torch = tc
    N, B, tile = 1 << 20, 128, 256
    chain = chain5(dspfx)
    chain[2] = dspfx.Reverb(seconds=0.5, decay=0.5)
    assert chain[2].delay_len == 24000
    chans = sample_channels(N)
    idx = torch.tensor(chans, device="cuda", dtype=torch.long)
    eng = dspfx.Engine(N, B, link_flags=3, tile_channels=tile)
    eng.set_chain(chain)
    s = torch.cuda.Stream()
    x = torch.empty(B * N, dtype=torch.float32, device="cuda")
    y = torch.empty_like(x)
    drag = [0.25, 0.75, 0.5, 0.75]
    after = [110, 300, 200, 300]                      # blocks after each store: past one period of the new ring
    total = 200 + sum(a + 128 for a in after)
    got_dev = torch.empty((total, B, len(chans)), dtype=torch.float32, device="cuda")
    busy_after, call_ms = [], []
    k = 0

    def block():
        nonlocal k
        eng.fill_noise(x, B, k * B, SEED, stream=s.cuda_stream)
        t0 = time.perf_counter()
        eng.process(x, out=y, n_frames=B, stream=s.cuda_stream)
        call_ms.append((time.perf_counter() - t0) * 1e3)
        busy_after.append(not s.query())
        with torch.cuda.stream(s):
            got_dev[k] = y.view(N // tile, B, tile)[idx // tile, :, idx % tile].transpose(0, 1)
        k += 1

    for _ in range(200):
        block()
    store_ms, seqs = [], []
    for seconds, n_after in zip(drag, after):
        go = threading.Event()

        def gui_thread():
            go.wait()
            t0 = time.perf_counter()
            seqs.append(eng.set_param_seq(2, 1, seconds))
            store_ms.append((time.perf_counter() - t0) * 1e3)

        t = threading.Thread(target=gui_thread)
        t.start()
        s.synchronize()
        _hold(torch, s, 120)
        for j in range(128):
            block()
            if j == 63:
                go.set()                               # 64 blocks queued, none has run: now the other thread moves the slider
        t.join()
        for _ in range(n_after):
            block()
    s.synchronize()
    assert k == total
    log = eng.param_log()
    assert [ev[0] for ev in log] == seqs and len(log) == 4
    stores = {}
    for (seq, frame, node, param, value), seconds in zip(log, drag):
        assert (node, param, value) == (2, 1, np.float32(seconds)) and frame % B == 0
        stores[frame // B] = seconds
    # the call that applied a store did not wait for the device: the stream it queued on was still busy when it returned
    # (the growth beyond capacity spends its time in the storing thread; by the time it is queued the hold may have run out)
    for i, blk in enumerate(sorted(stores)):
        if i != 1 and blk < total:
            assert busy_after[blk], (i, blk)
    landing = [call_ms[blk] for blk in sorted(stores) if blk < total]
    print("\nseconds drag at 1 048 576 channels: store calls (storing thread) %s ms; the process calls that applied them %s ms; slowest process "
          "call of all %.2f ms (block %d; median %.3f)" % (", ".join("%.2f" % m for m in store_ms), ", ".join("%.3f" % m for m in landing),
                                                          max(call_ms), int(np.argmax(call_ms)), float(np.median(call_ms))))
    assert max(landing) < 20.0, landing                # the call that swaps the ring in neither allocates nor waits for the device
    assert "36000 samples in 282 of 282 groups" in eng.describe(), eng.describe()      # capacity: the longest ring so far
    eng.close()
    got = got_dev.cpu().numpy().reshape(total * B, len(chans))
    descs = [n.oracle_desc() for n in chain]
    ref = np.empty_like(got)
    for i, c in enumerate(chans):
        xs = O.noise(SEED, np.array([c]), np.arange(total * B))[:, 0]
        nodes = [O.node_from_desc(d) for d in descs]
        for blk in range(total):
            if blk in stores:
                nodes[2].set_param(1, stores[blk])
            ref[blk * B:(blk + 1) * B, i] = O.chain_run(nodes, xs[blk * B:(blk + 1) * B], 3)
    assert ulp_diff(got, ref).max() <= 1, ulp_diff(got, ref).max()
    # the stores mattered: the echo of the 36000-sample ring is in the last phase
    last = max(stores)
    assert np.abs(ref[(last + 282) * B:]).max() > 0


def test_bench_py_prints_one_compact_contract_line(tmp_path):
    """bench.py end to end on the GPU (a reduced headline config so that it takes seconds): ONE JSON line on stdout that carries
    every contract field, is small enough for the driver's 8 KB tail, and agrees with itself (value = samples / time); the full
    record on stderr.  The line the driver parses must never break unnoticed."""
    import json
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "1", "--steps", "5", "--warmup", "2", "--channels", "65536", "--delay", "1024",
                        "--cpu-seconds", "1", "--paced-seconds", "0.2"], capture_output=True, text=True, timeout=900, cwd=root)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    assert len(lines[0]) < 7000
    d = json.loads(lines[0])
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data",
              "config", "roofline", "cpu_baseline"):
        assert k in d, k
    assert d["n_gpus"] == 1 and d["steps"] == 5 and d["warmup"] == 2 and d["dtype"] == "f32" and d["vs_baseline"] is None
    assert abs(d["value"] - 65536 * 128 / (d["ms_per_step"] * 1e-3)) / d["value"] < 1e-6
    rf = d["roofline"]
    assert rf["bound"] == "hbm" and abs(rf["frac"] - rf["achieved"] / rf["peak"]) < 1e-9 and 0.05 < rf["frac"] < 1.0 and rf["frac_by_step"] <= rf["frac"] * 1.001
    assert d["cpu_baseline"]["kind"] == "port" and d["cpu_baseline"]["value"] > 0 and d["paced"]["deadline_misses"] == 0
    assert "region_fixed_cost_us" in d and "workload" in d["config"]
    detail = [l for l in r.stderr.splitlines() if l.startswith("bench.py detail: ")]
    assert len(detail) == 1 and json.loads(detail[0][len("bench.py detail: "):])["value"] == d["value"]


def test_bare_gpus_2_command_starts_its_own_ranks_and_reports_both_exchange_forms():
    """`python3 bench.py --gpus 2 ...` with NO torch.distributed.run in front of it and no launcher environment --
    the shape of the driver's recorded N = 1 command -- must start its own ranks (fresh children, before any GPU call of the
    parent), exchange the bus of every block between them, and print ONE line with both forms of the exchange.  On this one-GPU
    box both ranks share device 0 (DSPFX_BENCH_SHARE_GPU=1: gloo for torch's group, the C ABI's mailbox communicator for the bus);
    a reduced shard so that it takes seconds.  Without that switch the same command must refuse to run on one GPU -- never a silent
    N = 1."""
    import json
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "LOCAL_WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT",
                                                            "GROUP_RANK", "ROLE_RANK", "TORCHELASTIC_RUN_ID")}
    cmd = [sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "20", "--warmup", "5", "--channels", "65536", "--delay", "1024"]
    import torch
    if torch.cuda.device_count() < 2:
        r = subprocess.run(cmd, capture_output=True, text=True, timeout=300, cwd=root, env=env)
        assert r.returncode != 0 and r.stdout.strip() == "" and "needs 2 visible GPU(s)" in r.stderr, r.stderr[-2000:]
    env.update(DSPFX_BENCH_SHARE_GPU="1", DSPFX_BENCH_COMM="abi", HSA_ENABLE_IPC_MODE_LEGACY="0")
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=900, cwd=root, env=env)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["steps"] == 20 and d["scaling"] == "weak" and d["cpu_baseline"] is None
    assert d["config"]["parallelism"] == "channel-shard x2" and d["collective_backend"] == "mailbox" and d["collective_fallback"] is None
    assert abs(d["value"] - 2 * 65536 * 128 / (d["ms_per_step"] * 1e-3)) / d["value"] < 1e-6
    f = d["scaling_forms"]
    assert f["inline"]["value"] == pytest.approx(d["value"], rel=1e-6) and f["inline"]["bus_delay_blocks"] == 0
    assert f["overlapped"]["value"] > 0 and f["overlapped"]["bus_delay_blocks"] == 2, f
    assert f["same_block_second_stream"]["value"] > 0 and f["same_block_second_stream"]["bus_delay_blocks"] == 0, f
    assert d["bus_exchange"]["backend"] == "mailbox" and 0 < d["bus_exchange"]["us_p50"] < 1000
    assert "starting 2 ranks" in r.stderr
    print("bare --gpus 2 on one GPU: inline %.3g samples/s, overlapped %.3g, exchange p50 %.1f us" % (
        f["inline"]["value"], f["overlapped"]["value"], d["bus_exchange"]["us_p50"]))
