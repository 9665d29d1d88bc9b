#!/usr/bin/env python3
"""Delay-ring length changes in mid-stream on random chains, under the product's defaults (round 5: a length change is three
scalars, groups are appended and never zeroed, the storing thread allocates).  Seeded random chains of the exact-arithmetic kinds
with one to three Reverb nodes that carry a seconds slider (restored, menu-fresh under either reading of rivulet, or an explicit
ring); ragged and whole channel counts, both layouts, blocks of 100 / 128 / 256 frames; six to ten actions per run at random
blocks -- a `seconds` store (144 .. 2400 samples), a `decay` store (the same ring length again), dspfx_set_delay_len, a
dspfx_reserve_delay_len hint, dspfx_ring_trim, dspfx_reset -- half of the runs make their stores from a SECOND THREAD while
blocks are queued behind a held stream (the oracle then takes them at the blocks the engine logged).  Every output sample
against the oracle: ulp and the sign of zeros.
usage: r05_ring_soak.py [first_seed] [count]"""
import os, sys, threading, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests"), os.path.join(ROOT, "oracle")):
    sys.path.insert(0, p)
import numpy as np
import torch
from __graft_entry__ import load_package
E = load_package()
import oracle as O
import test_gpu_parity as T
from chains import ulp_diff
from test_gpu_threads import _hold


def random_reverb(rng):
    k = int(rng.integers(0, 4))
    decay = float(rng.uniform(0.0, 0.9))
    if k == 0:
        return E.Reverb(seconds=float(rng.uniform(0.003, 0.03)), decay=decay, page_round=bool(rng.integers(0, 2)))
    if k == 1:
        n = E.Reverb(page_round=bool(rng.integers(0, 2)))                  # fresh from the menu: 128 / 1024 samples under a slider ...
        n.params[0] = decay
        n.params[1] = float(rng.uniform(0.003, 0.03))                      # ... that asks for something else at the first touch
        return n
    return E.Reverb(delay_samples=int(rng.integers(128, 900)), decay=decay)


def run(s0=5000, cnt=60, budget_s=None):
    """see the module docstring; budget_s stops the sweep early (the test suite's time box); returns the counters."""
    t0, worst, bad, ran, n_actions, threaded = time.time(), 0, [], 0, 0, 0
    for seed in range(s0, s0 + cnt):
        if budget_s is not None and time.time() - t0 > budget_s:
            break
        rng = np.random.default_rng(seed)
        chain = []
        for _ in range(int(rng.integers(1, 8))):
            n = T._random_exact_node(E, rng)
            if n.kind not in (E.ADD, E.MIX, E.REVERB):
                chain.append(n)
        for _ in range(int(rng.integers(1, 4))):
            chain.insert(int(rng.integers(0, len(chain) + 1)), random_reverb(rng))
        chain = chain[:10]
        revs = [i for i, n in enumerate(chain) if n.kind == E.REVERB]
        tile = int(rng.choice([0, 0, 64, 256]))
        N = int(rng.choice([256, 1024, 4096])) if tile else int(rng.choice([1, 63, 100, 273, 1000, 2085, 576]))
        block = int(rng.choice([128, 128, 256, 100]))
        lf = int(rng.choice([0, 1, 3]))
        nblocks = 36
        x = T.noise_block(N, block * nblocks, seed=seed)
        acts = {}
        for _ in range(int(rng.integers(6, 11))):
            i = int(rng.choice(revs))
            what = int(rng.integers(0, 7))
            k = int(rng.integers(1, nblocks))
            if what <= 2:
                a = ("param", i, 1, float(np.float32(rng.uniform(0.003, 0.05))))
            elif what == 3:
                a = ("param", i, 0, float(np.float32(rng.uniform(0.0, 0.9))))
            elif what == 4:
                a = ("len", i, int(rng.integers(128, 2400)))
            elif what == 5:
                a = ("reserve", i, int(rng.integers(128, 4000)))
            else:
                a = ("trim",) if rng.random() < 0.5 else ("reset",)
            acts.setdefault(k, []).append(a)
        use_thread = bool(rng.integers(0, 2))
        eng = E.Engine(N, block, link_flags=lf, tile_channels=tile)
        eng.set_chain(chain)
        dx = torch.from_numpy(np.concatenate([E.to_layout(x[k * block:(k + 1) * block], tile).reshape(-1) for k in range(nblocks)])).cuda()
        dy = torch.empty_like(dx)
        s = torch.cuda.Stream()
        applied = {}                          # block -> actions as they really landed (threaded stores: from the engine's log)
        per = block * N
        pending_threads = []

        def do(a, blk):
            if a[0] == "param":
                return eng.set_param_seq(a[1], a[2], a[3])
            if a[0] == "len":
                eng.set_delay_len(a[1], a[2])
            elif a[0] == "reserve":
                eng.reserve_delay_len(a[1], a[2])
            elif a[0] == "trim":
                eng.ring_trim()
            elif a[0] == "reset":
                eng.reset()
            return None

        seq_of, direct = {}, set()
        torch.cuda.synchronize()
        for k in range(nblocks):
            todo = acts.get(k, [])
            n_actions += len(todo)
            if use_thread and todo and all(a[0] in ("param", "reserve") for a in todo):
                threaded += 1
                _hold(torch, s, 15)
                go = threading.Event()

                def gui(todo=todo, k=k):
                    go.wait()
                    for a in todo:
                        q = do(a, k)
                        if q is not None:
                            seq_of[q] = a
                th = threading.Thread(target=gui)
                th.start()
                pending_threads.append((th, go))
                go.set()
            else:                             # made here, between two process calls: they take effect at once, in call order
                if todo:
                    for th, _ in pending_threads:          # (stores still on their way from the other thread land first: keeps the replay order simple)
                        th.join()
                for a in todo:
                    q = do(a, k)
                    if q is not None:
                        direct.add(q)
                    if a[0] in ("param", "len", "reset"):
                        applied.setdefault(k, []).append((len(applied.get(k, [])) - 1000, a))
            eng.process(dx[k * per:(k + 1) * per], out=dy[k * per:(k + 1) * per], n_frames=block, stream=s.cuda_stream)
        for th, _ in pending_threads:
            th.join()
        s.synchronize()
        log_copy = eng.param_log()
        for seq, frame, node, param, value in log_copy:
            if seq not in direct:             # a store from the second thread: it landed where the engine's log says -- and BEFORE that block's
                # direct actions: the thread is joined before those are made, and a store still queued then is drained on entry of the first of them
                applied.setdefault(frame // block, []).append((seq - 10 ** 6, ("param", node, param, value)))
        y = np.concatenate([E.from_layout(dy[k * per:(k + 1) * per].cpu().numpy(), block, N, tile) for k in range(nblocks)])
        eng.close()
        descs = [n.oracle_desc() for n in chain]
        chans = range(N) if N <= 300 else sorted(set(rng.integers(0, N, 96).tolist()) | {0, N - 1})
        ref = np.empty((block * nblocks, len(chans)), np.float32)
        for j, c in enumerate(chans):
            nodes = [O.node_from_desc(d) for d in descs]
            for k in range(nblocks):
                for seq, a in sorted(applied.get(k, []), key=lambda t: t[0]):
                    if a[0] == "param":
                        nodes[a[1]].set_param(a[2], a[3])
                    elif a[0] == "len":
                        nodes[a[1]].set_delay_len(a[2])
                    elif a[0] == "reset":
                        for nd in nodes:
                            nd.reset()
                ref[k * block:(k + 1) * block, j] = O.chain_run(nodes, x[k * block:(k + 1) * block, c], lf, block=min(block, 128))
        got = y[:, list(chans)]
        ran += 1
        ok = np.isfinite(ref)
        if not np.array_equal(np.isfinite(got), ok):
            bad.append((seed, "finite"))
            continue
        d = ulp_diff(got[ok], ref[ok])
        w = int(d.max()) if d.size else 0
        worst = max(worst, w)
        if w > 1 or not np.array_equal(np.signbit(got[ok]), np.signbit(ref[ok])):
            bad.append((seed, w, N, tile, block, lf, use_thread, [n.kind for n in chain]))
            if os.environ.get("DSPFX_SOAK_VERBOSE"):
                dd = ulp_diff(got, ref)
                first = int(np.argwhere(dd.max(axis=1) > 1)[0][0])
                print("seed", seed, "first bad frame", first, "= block", first // block, "+", first % block, "chain", [(n.kind, list(n.params[:2]), n.mode, n.delay_len) for n in chain])
                print("  acts", {k: v for k, v in sorted(acts.items())})
                print("  applied", {k: v for k, v in sorted(applied.items())})
                print("  log", log_copy)
        if (seed - s0) % 10 == 9:
            print("... %d runs, worst %d ulp, failures %s, %.0f s" % (ran, worst, bad, time.time() - t0), flush=True)
    print("seeds %d..%d: %d runs, %d actions (%d blocks with stores from a second thread); worst ulp vs oracle %d, failures %s, %.0f s" % (
        s0, s0 + cnt - 1, ran, n_actions, threaded, worst, bad, time.time() - t0))
    return dict(ran=ran, actions=n_actions, threaded_blocks=threaded, worst=worst, bad=bad, seconds=time.time() - t0)


if __name__ == "__main__":
    run(int(sys.argv[1]) if len(sys.argv) > 1 else 5000, int(sys.argv[2]) if len(sys.argv) > 2 else 60)
