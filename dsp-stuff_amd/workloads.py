"""The synthetic workloads of BASELINE.json's configs: the chains, their slider values and
config 4's taps.  bench.py times them, the parity tests check them against the oracle; both import them from here
(tests/chains.py re-exports these names), so the benchmark never reaches into tests/.

`pkg` is the loaded package (`__graft_entry__.load_package()`): the node constructors carry the reference's
names and slider fields (dsp-stuff/src/nodes/*.rs)."""
import math

import numpy as np


def rbj_lowpass(fc, q, fs=48000.0):
    """RBJ cookbook low-pass; computed in f64, handed over as the six raw sliders of nodes/biquad.rs:20-37."""
    w0 = 2 * math.pi * fc / fs
    al = math.sin(w0) / (2 * q)
    c = math.cos(w0)
    return [1 + al, -2 * c, 1 - al, (1 - c) / 2, 1 - c, (1 - c) / 2]   # a0,a1,a2,b0,b1,b2


def rbj_highpass(fc, q, fs=48000.0):
    w0 = 2 * math.pi * fc / fs
    al = math.sin(w0) / (2 * q)
    c = math.cos(w0)
    return [1 + al, -2 * c, 1 - al, (1 + c) / 2, -(1 + c), (1 + c) / 2]


def chain3(pkg, delay=24000):
    """BASELINE configs 1/2: gain(0.8) -> biquad LP 1 kHz -> delay(D, 0.5)"""
    return [pkg.Gain(0.8), pkg.BiQuad(*rbj_lowpass(1000.0, 0.7071)), pkg.Reverb(delay_samples=delay, decay=0.5)]


def chain5(pkg, delay=24000):
    """BASELINE configs 3/5: biquad LP 1k -> SoftClip(3) -> delay(D,0.5) -> biquad HP 80 -> gain(0.5)"""
    return [pkg.BiQuad(*rbj_lowpass(1000.0, 0.7071)), pkg.Distort(3.0, pkg.SOFT_CLIP),
            pkg.Reverb(delay_samples=delay, decay=0.5), pkg.BiQuad(*rbj_highpass(80.0, 0.7071)), pkg.Gain(0.5)]


def fir_taps(T, seed=0x5EED0004):
    """Config 4: h[j] = u_j * exp(-6.9 j / T), f64"""
    rng = np.random.default_rng(seed)
    return rng.uniform(-1.0, 1.0, T) * np.exp(-6.9 * np.arange(T) / T)
