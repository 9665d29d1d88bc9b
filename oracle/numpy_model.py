"""Independent numpy-float32 model of the reference effect arithmetic.

TEST INFRASTRUCTURE ONLY -- PARITY UNPINNED (see dspfx_oracle.h).  Written
separately from the C oracle (different language, vectorised where the maths
allows) so the two restatements can cross-check each other and generate the
golden vectors in tests/golden/.  Citations relative to /root/reference/.
"""
from __future__ import annotations

import numpy as np

F = np.float32
BUF_SIZE = 128  # dsp-stuff/src/node.rs:257


def link_scale(x):
    """node.rs:162-194 with ONE connected pipe: (0 + x) / f32(0.0001 + 1)."""
    div = F(F(0.0001) + F(1.0))
    return ((F(0.0) + np.asarray(x, F)) / div).astype(F)


def slider_input(ctl, lo, hi):
    """dsp-stuff-derive/src/lib.rs:141-146"""
    y = (np.asarray(ctl, F) + F(1.0)) / F(2.0)
    z = np.where(y < F(0), F(0), y)
    z = np.where(z > F(1), F(1), z).astype(F)
    return (F(lo) + (F(hi) - F(lo)) * z).astype(F)


def gain(x, level):
    """gain.rs:33-37"""
    return (np.asarray(x, F) * F(level)).astype(F)


class Biquad:
    """biquad.rs:48-88 + biquad 0.4.2 DirectForm1 (restated)."""

    def __init__(self, a0=1.0, a1=-0.24, a2=0.0, b0=0.758, b1=0.0, b2=0.0, normalise=True):
        a0 = F(a0)
        if normalise:  # regenerate_filter, biquad.rs:62-76
            self.a1, self.a2 = F(a1) / a0, F(a2) / a0
            self.b0, self.b1, self.b2 = F(b0) / a0, F(b1) / a0, F(b2) / a0
        else:          # initial_filter, biquad.rs:48-60
            self.a1, self.a2, self.b0, self.b1, self.b2 = F(a1), F(a2), F(b0), F(b1), F(b2)
        self.x1 = self.x2 = self.y1 = self.y2 = F(0)

    def run(self, x):
        out = np.empty(len(x), F)
        for i, v in enumerate(np.asarray(x, F)):
            y = F(F(F(F(self.b0 * v) + F(self.b1 * self.x1)) + F(self.b2 * self.x2))
                  - F(self.a1 * self.y1)) - F(self.a2 * self.y2)
            y = F(y)
            self.x2, self.x1, self.y2, self.y1 = self.x1, v, self.y1, y
            out[i] = y
        return out


class OnePole:
    def __init__(self, ratio=0.5, high=False):
        self.r, self.z, self.high = F(ratio), F(0), high

    def run(self, x):
        out = np.empty(len(x), F)
        one_m = F(F(1.0) - self.r)
        for i, v in enumerate(np.asarray(x, F)):
            t = F(F(v * one_m) + F(self.r * self.z))
            if self.high:   # high_pass.rs:36-39
                self.z = t
                out[i] = F(v - t)
            else:           # low_pass.rs:36-39
                out[i] = t
                self.z = t
        return out


class Reverb:
    """reverb.rs:76-110 as y[n] = x[n] + decay*y[n-D], zero history."""

    def __init__(self, d, decay=0.5):
        self.hist = np.zeros(int(d), F)
        self.d, self.decay, self.n = int(d), F(decay), 0

    def run(self, x):
        out = np.empty(len(x), F)
        for i, v in enumerate(np.asarray(x, F)):
            p = (self.n + i) % self.d
            y = F(v + F(self.hist[p] * self.decay))
            self.hist[p] = y
            out[i] = y
        self.n += len(x)
        return out


def _clip(s):
    s = np.asarray(s, F)
    return np.where(s < F(-1), F(-1), np.where(s > F(1), F(1), s)).astype(F)


def _signum(x):
    x = np.asarray(x, F)
    s = np.where(np.signbit(x), F(-1), F(1)).astype(F)
    return np.where(np.isnan(x), x, s).astype(F)


def distort(x, level, mode):
    """distort.rs:63-172 (one reference block for Fuzz)."""
    x = np.asarray(x, F)
    L = F(level)
    if mode != 4 and L < F(0.001):
        return x.copy()
    with np.errstate(all="ignore"):
        if mode == 0:
            return (_clip(x * L) / L).astype(F)
        if mode == 1:
            s = (x * L).astype(F)
            cube = ((s * s).astype(F) * s).astype(F)
            mid = (s - (cube / F(3.0)).astype(F)).astype(F)
            t = np.where(s > F(1), F(2.0) / F(3.0),
                         np.where((s >= F(-1)) & (s <= F(1)), mid, F(-2.0) / F(3.0))).astype(F)
            return (_clip(t) / L).astype(F)
        if mode == 2:
            return np.tanh((x * L).astype(F)).astype(F)
        if mode == 3:
            den = ((np.abs(x) * L).astype(F) + F(1)).astype(F)
            return (_signum(x) * (F(1) - (F(1) / den).astype(F)).astype(F)).astype(F)
        if mode == 4:
            lv = np.full(len(x), L, F)
            mx = np.abs(x).max() if not np.isnan(x).any() else F(np.nan)
            q = (_clip(x * lv) / mx).astype(F)
            z = (-np.abs(F(1) - np.exp(-np.abs(q)).astype(F))).astype(F)
            mz = np.abs(z).max() if not np.isnan(z).any() else F(np.nan)
            y = (_clip((z * mx).astype(F)) / mz).astype(F)
            my = np.abs(y).max() if not np.isnan(y).any() else F(np.nan)
            return ((y * mx).astype(F) / my).astype(F)
        if mode == 5:
            return np.sin((x * L).astype(F)).astype(F)
        if mode == 6:
            return np.arctan((x * L).astype(F)).astype(F)
        if mode == 7:
            v = (x * L).astype(F)
            return ((v * v).astype(F) * _signum(v)).astype(F)
        if mode == 8:
            v = (x * L).astype(F)
            v2 = (v * v).astype(F)
            v4 = (v2 * v2).astype(F)
            return (((F(8) * v4).astype(F) - (F(8) * v2).astype(F)).astype(F) + F(1)).astype(F)
    raise ValueError(mode)


def overdrive(x, boost, drive, level):
    """overdrive.rs:31-43"""
    x = np.asarray(x, F)
    if F(level) < F(0.001):
        return x.copy()
    a = (x * F(boost)).astype(F)
    b = (F(np.pi / 4) * a).astype(F)
    c = np.arctan(b).astype(F)
    d = (F(2 / np.pi) * c).astype(F)
    mix = ((F(drive) * d).astype(F) + (F(F(1) - F(drive)) * x).astype(F)).astype(F)
    return (mix * F(level)).astype(F)


def chebyshev(x, level_pos, level_neg):
    """chebyshev.rs:28-42"""
    x = np.asarray(x, F)
    out = x.copy()
    lp, ln = F(level_pos), F(level_neg)
    pos = x >= F(0)
    if lp >= F(0.001):
        out[pos] = (np.tanh((x[pos] * lp).astype(F)).astype(F) / np.tanh(lp)).astype(F)
    neg = ~pos
    if ln >= F(0.001):
        out[neg] = (np.tanh((x[neg] * ln).astype(F)).astype(F) / np.tanh(ln)).astype(F)
    return out


class Fir:
    """fir.rs:180-224, mathematically (f64 accumulate, single slice): warm-up pairs
    state[k] with taps_rev[k]; steady state is causal convolution."""

    def __init__(self, taps_reversed, average=False):
        self.t = np.asarray(taps_reversed, np.float64)
        self.state = np.zeros(0, np.float64)
        self.div = F(1.0) / F(len(self.t)) if average else F(1.0)

    def run(self, x):
        out = np.empty(len(x), F)
        T = len(self.t)
        for i, v in enumerate(np.asarray(x, F)):
            self.state = np.append(self.state, np.float64(v))
            if len(self.state) > T:
                self.state = self.state[1:]
            L = len(self.state)
            acc = np.float64(0)
            for k in range(L):  # sequential f64 sum like Iterator::sum
                acc += self.state[k] * self.t[k]
            out[i] = F(F(acc) * self.div)
        return out


def add(a, b):
    return (np.asarray(a, F) + np.asarray(b, F)).astype(F)


def mix(a, b, ratio):
    """mix.rs:41-46"""
    r = F(ratio)
    return ((np.asarray(b, F) * r).astype(F) + (np.asarray(a, F) * F(F(1) - r)).astype(F)).astype(F)


class SignalGen:
    """signal_gen.rs:57-129, one channel.  mode: 0 Sine, 1 Triangle, 2 Square, 3 Constant.
    float32 sin is taken as the correctly rounded value (numpy's float64 sin, rounded once)."""

    TAU = F(6.283185307179586)

    def __init__(self, amplitude=0.5, frequency=100.0, mode=0):
        self.amplitude, self.frequency, self.mode = F(amplitude), F(frequency), mode
        self.clock = F(0.0)

    def process(self, n=BUF_SIZE, amplitude=None, frequency=None):
        """One block of n frames; amplitude / frequency are optional per-sample (control port) arrays."""
        amp = np.full(n, self.amplitude, F) if amplitude is None else np.asarray(amplitude, F)
        frq = np.full(n, self.frequency, F) if frequency is None else np.asarray(frequency, F)
        if self.mode == 3:
            return amp.copy()
        step = (frq / F(48000.0)).astype(F)
        total = np.empty(n, F)
        acc = F(0.0)
        for i in range(n):            # sequential f32 accumulation (signal_gen.rs:65)
            acc = F(acc + step[i])
            total[i] = acc
        ph = (self.clock + total).astype(F)
        if self.mode == 0:
            out = (np.sin((ph * self.TAU).astype(F).astype(np.float64)).astype(F) * amp).astype(F)
        elif self.mode == 1:
            out = ((F(2.0) * np.fmod(ph, F(1.0)).astype(F) - F(1.0)).astype(F) * amp).astype(F)
        else:
            out = (np.where(total > F(0.5), F(1.0), F(-1.0)).astype(F) * amp).astype(F)
        self.clock = F(np.fmod(F(self.clock + acc), F(1.0)))
        return out


class Envelope:
    """envelope.rs:34-52 over dasp_envelope 0.11.0 Detector<f32, Peak<FullWave>> (restated as recalled, see
    dspfx_oracle.h): env = d + (env - d) * gain, d = |x|, gain = attack gain while env < d else release gain."""

    E = F(2.718281828459045)

    def __init__(self, attack=0.0, release=0.0):
        self.attack, self.release = F(attack), F(release)
        self.env = F(0.0)

    @classmethod
    def calc_gain(cls, n_frames):
        n = F(n_frames)
        if n == F(0):
            return F(0.0)
        # powf(e_f32, x): evaluated in float64 and rounded once (glibc powf is within 1 ulp of that)
        return F(np.power(np.float64(cls.E), np.float64(F(-1.0) / n)))

    def process(self, x):
        x = np.asarray(x, F)
        ga, gr = self.calc_gain(self.attack), self.calc_gain(self.release)
        out = np.empty_like(x)
        l = self.env
        for i in range(x.size):
            d = -x[i] if x[i] < F(0) else x[i]
            g = ga if l < d else gr
            l = F(d + F(F(l + (-d)) * g))
            out[i] = l
        self.env = l
        return out
