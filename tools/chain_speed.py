#!/usr/bin/env python3
"""Steady-state step time of one chain on one engine, placement-tuned against its buffers (reproducible to a few
tenths of a percent), for comparing kernel variants:  DSPFX_VARIANT="static=0,f=8,cpl=2" python tools/chain_speed.py tanh
chains: chain5 | tanh (biquad > tanh > delay > biquad > gain) | eight (8 arithmetic nodes) | onepole"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
from __graft_entry__ import load_package
import chains
pkg = load_package()
name = sys.argv[1] if len(sys.argv) > 1 else "chain5"
tile = int(sys.argv[2]) if len(sys.argv) > 2 else 256
N, B, D = 1 << 20, 128, 24000
lp, hp = chains.rbj_lowpass(1000.0, 0.7071), chains.rbj_highpass(80.0, 0.7071)
chain = {
    "chain5": lambda: chains.chain5(pkg, D),
    "tanh": lambda: [pkg.BiQuad(*lp), pkg.Distort(3.0, pkg.TANH), pkg.Reverb(delay_samples=D, decay=0.5), pkg.BiQuad(*hp), pkg.Gain(0.5)],
    "eight": lambda: [pkg.Gain(0.9), pkg.BiQuad(*lp), pkg.LowPass(0.3), pkg.Distort(2.0, pkg.HARD_CLIP), pkg.Reverb(delay_samples=D, decay=0.4),
                      pkg.HighPass(0.1), pkg.Envelope(5.0, 300.0), pkg.Gain(1.1)],
    "onepole": lambda: [pkg.LowPass(0.3), pkg.Gain(0.5)],
    "d_tanh": lambda: [pkg.Distort(3.0, pkg.TANH)], "d_sin": lambda: [pkg.Distort(3.0, pkg.SIN)], "d_atan": lambda: [pkg.Distort(3.0, pkg.ATAN)],
    "overdrive": lambda: [pkg.Overdrive(5.0, 0.5, 0.8)], "chebyshev": lambda: [pkg.Chebyshev(4.0, 2.0)],
    "fuzz": lambda: [pkg.Distort(3.0, pkg.FUZZ)], "fuzz3": lambda: [pkg.BiQuad(*lp), pkg.Distort(3.0, pkg.FUZZ), pkg.Gain(0.5)],
    "ctl": lambda: [pkg.Gain(1.0), pkg.BiQuad(*lp)],
    "siggen": lambda: [pkg.SignalGen(0.5, 440.0, pkg.SIG_SINE)], "gain": lambda: [pkg.Gain(0.5)],
}[name]()
eng = pkg.Engine(N, B, link_flags=3, tile_channels=tile)
eng.set_chain(chain)
x = torch.empty(B * N, dtype=torch.float32, device="cuda"); y = torch.empty_like(x)
eng.fill_noise(x, B, 0)
ctl = None
if name == "ctl":          # a control port on the gain's level: +4 B/sample
    cbuf = torch.empty_like(x); eng.fill_noise(cbuf, B, 7)
    ctl = {(0, 0): cbuf}
eng.tune_placement(x, y, B)
def measure(steps=192):
    for _ in range(200): eng.process(x, out=y, n_frames=B, ctl=ctl)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(steps): eng.process(x, out=y, n_frames=B, ctl=ctl)
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / steps
ms = min(measure(), measure())
bps = eng.algorithmic_bytes_per_sample(B) + (4 if ctl else 0)
stage = [l for l in eng.describe().splitlines() if l.startswith("stage")]
print("%-8s tile %3d  %.4f ms/step  %5.0f GB/s  %s" % (name, tile, ms, bps * N * B / ms / 1e6, " | ".join(s.split(":")[1].split("(")[0].strip() for s in stage)))
