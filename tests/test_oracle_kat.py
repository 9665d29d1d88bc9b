"""Known-answer tests pinning the CPU oracle to the reference source lines
(SURVEY.md 8c KAT-1..10).  The reference ships no tests (PARITY UNPINNED), so
every expectation here is derived from the cited lines by hand / closed form,
and cross-checked against the independent numpy-float32 model."""
import math

import numpy as np
import pytest

import numpy_model as M
import oracle as O

F = np.float32


def bits(a):
    return np.asarray(a, F).view(np.uint32)


def test_kat1_gain_bit_exact():
    # gain.rs:33-37  out = in * level
    rng = np.random.default_rng(1)
    x = rng.standard_normal(128).astype(F)
    x[:6] = [0.0, -0.0, 1e-42, -1e-42, np.inf, -np.inf]
    for level in (0.0, 0.8, 1.0, 10.0, 3.3333):
        n = O.Node(O.GAIN, [level])
        y = n.process(x)
        assert np.array_equal(bits(y), bits(x * F(level)))
        assert np.array_equal(bits(y), bits(M.gain(x, level)))
    assert math.isnan(O.Node(O.GAIN, [2.0]).process(np.array([np.nan], F))[0])


def test_kat2_link_scale():
    # node.rs:166,179,190: one pipe => divisor f32(0.0001 + 1.0) = 0x1.00068ep+0
    div = O.link_divisor(1)
    assert float(div).hex() == "0x1.00068e0000000p+0"
    n = O.Node(O.GAIN, [1.0])
    y = O.chain_run([n], np.array([0.5, -0.0, 1.0], F), link_flags=O.LINK_INPUT)
    assert float(y[0]).hex() == "0x1.fff2e40000000p-2"
    assert bits(y)[1] == 0  # 0.0 + (-0.0) = +0.0
    # k hops = k successive divisions
    nodes = [O.Node(O.GAIN, [1.0]) for _ in range(5)]
    y5 = O.chain_run(nodes, np.array([0.5], F), link_flags=3)
    e = F(0.5)
    for _ in range(5):
        e = F(e / div)
    assert bits(y5)[0] == bits(e)
    # internal-only: 4 hops for 5 nodes
    nodes = [O.Node(O.GAIN, [1.0]) for _ in range(5)]
    y4 = O.chain_run(nodes, np.array([0.5], F), link_flags=O.LINK_INTERNAL)
    e = F(0.5)
    for _ in range(4):
        e = F(e / div)
    assert bits(y4)[0] == bits(e)
    # mix-bus divisor: sequential f32 increments (node.rs:179)
    assert O.link_divisor(0) == F(0.0001)
    assert O.link_divisor(2) == F(F(F(0.0001) + F(1)) + F(1))
    assert O.link_divisor(1 << 20) == F(1 << 20)


def test_kat3_biquad_defaults_impulse():
    # biquad.rs:48-60: a1=-0.24, b0=0.758 => y[n] = 0.758*x[n] + 0.24*y[n-1]
    n = O.Node(O.BIQUAD)
    x = np.zeros(64, F)
    x[0] = 1
    y = n.process(x)
    e = np.empty(64, F)
    prev = F(0)
    for i in range(64):
        prev = F(F(F(F(F(0.758) * x[i]) + F(0)) + F(0)) - F(F(-0.24) * prev)) - F(0)
        e[i] = prev
    assert np.array_equal(bits(y), bits(e))
    assert abs(float(y[5]) - 0.758 * 0.24 ** 5) < 1e-7
    # state reset on any param change (biquad.rs:74)
    n.set_param(3, 0.758)
    y2 = n.process(x)
    assert np.array_equal(bits(y2), bits(y))


def test_kat4_biquad_general_vs_numpy_and_scipy():
    from scipy.signal import lfilter
    rng = np.random.default_rng(4)
    x = rng.uniform(-1, 1, 128).astype(F)
    for _ in range(8):
        # stable poles inside unit circle
        r, th = rng.uniform(0.1, 0.95), rng.uniform(0.1, 3.0)
        a0 = rng.uniform(0.5, 2.0)
        a = np.array([1.0, -2 * r * math.cos(th), r * r]) * a0
        b = rng.uniform(-1, 1, 3) * a0
        p = [a[0], a[1], a[2], b[0], b[1], b[2]]
        n = O.Node(O.BIQUAD, p)
        y = n.process(x)
        m = M.Biquad(*p)
        assert np.array_equal(bits(y), bits(m.run(x)))
        ref = lfilter(np.array(b, F).astype(np.float64) / F(a[0]), np.array(a, F).astype(np.float64) / F(a[0]), x.astype(np.float64))
        assert np.max(np.abs(ref - y)) < 5e-5


def test_kat5_one_pole():
    x = np.ones(128, F)
    for r in (0.0, 0.5, 0.9, 1.0):
        lp = O.Node(O.LOW_PASS, [r]).process(x)
        hp = O.Node(O.HIGH_PASS, [r]).process(x)
        assert np.array_equal(bits(lp), bits(M.OnePole(r).run(x)))
        assert np.array_equal(bits(hp), bits(M.OnePole(r, high=True).run(x)))
        k = np.arange(1, 129)
        assert np.allclose(lp, 1 - r ** k, atol=1e-5)       # step response 1 - r^(n+1)
        assert np.allclose(hp, r ** k, atol=1e-5)
    # state carries across blocks (low_pass.rs:34,41)
    n = O.Node(O.LOW_PASS, [0.5])
    a = n.process(x[:64])
    b = n.process(x[:64])
    assert np.array_equal(np.concatenate([a, b]), O.Node(O.LOW_PASS, [0.5]).process(x))


@pytest.mark.parametrize("D", [128, 1024, 24000])
def test_kat6_delay_impulse(D):
    # reverb.rs:86-103: y[n] = x[n] + decay*y[n-D]
    nblk = (3 * D) // 128 + 2
    x = np.zeros(nblk * 128, F)
    x[0] = 1
    n = O.Node(O.REVERB, [0.5], delay_len=D)
    y = O.chain_run([n], x, link_flags=0)
    e = np.zeros_like(x)
    for k in range(len(e) // D + 1):
        if k * D < len(e):
            e[k * D] = 0.5 ** k
    assert np.array_equal(bits(y), bits(e))
    m = M.Reverb(D, 0.5)
    assert np.array_equal(bits(m.run(x)), bits(e))


def test_kat6_delay_block_invariance():
    rng = np.random.default_rng(6)
    x = rng.uniform(-1, 1, 1024).astype(F)
    ys = []
    for blk in (128, 64, 32):
        n = O.Node(O.REVERB, [0.7], delay_len=128)
        ys.append(O.chain_run([n], x, link_flags=0, block=blk))
    assert np.array_equal(ys[0], ys[1]) and np.array_equal(ys[0], ys[2])


def test_kat6_delay_len_helper():
    # reverb.rs:58
    assert O.delay_len(0.5) == 24000
    assert O.delay_len(0.5, True) == 24576
    assert O.delay_len(0.0) == 128
    assert O.delay_len(1.0) == 48000
    assert O.delay_len(0.001) == 128
    # default node keeps make_buffer()'s 128-sample ring (reverb.rs:44-52)
    n = O.Node(O.REVERB)
    x = np.zeros(256, F)
    x[0] = 1
    y = O.chain_run([n], x, link_flags=0)
    assert y[128] == F(0.5) and y[0] == 1


def test_kat7_distort_tables():
    xs = np.array([0, -0.0, 0.25, -0.25, 0.5, -0.5, 1, -1, 2, -2], F)
    two_thirds = F(2.0) / F(3.0)
    assert float(two_thirds).hex() == "0x1.5555560000000p-1"
    for mode in (0, 1, 2, 3, 5, 6, 7, 8):
        for L in (0.0, 0.0009, 0.001, 1.0, 3.0, 30.0):
            y = O.Node(O.DISTORT, [L], mode=mode).process(xs)
            m = M.distort(xs, L, mode)
            if F(L) < F(0.001):   # bypass threshold (distort.rs:64 etc.)
                assert np.array_equal(bits(y), bits(xs))
                continue
            if mode in (2, 5, 6):  # libm: glibc vs numpy may differ by an ulp
                assert np.allclose(y, m, rtol=3e-7, atol=1e-7)
            else:
                assert np.array_equal(bits(y), bits(m)), (mode, L)
    # SoftClip plateaus +-(2/3)/L, HardClip saturation
    y = O.Node(O.DISTORT, [3.0], mode=O.SOFT_CLIP).process(np.array([2, -2], F))
    assert y[0] == F(two_thirds / F(3)) and y[1] == F(-two_thirds / F(3))
    y = O.Node(O.DISTORT, [3.0], mode=O.HARD_CLIP).process(np.array([2, -2, 0.1], F))
    assert y[0] == F(F(1) / F(3)) and y[1] == F(F(-1) / F(3)) and y[2] == F(F(F(0.1) * F(3)) / F(3))
    # Chebyshev4(0) = 1 ; signum(+-0) = +-1
    assert O.Node(O.DISTORT, [1.0], mode=O.CHEBYSHEV4).process(np.array([0], F))[0] == 1
    y = O.Node(O.DISTORT, [1.0], mode=O.RECIP_SOFT_CLIP).process(np.array([0.0, -0.0], F))
    assert bits(y)[0] == 0 and bits(y)[1] == 0x80000000   # +-1 * (1 - 1/1) = +-0
    y = O.Node(O.DISTORT, [1.0], mode=O.SQUARE).process(np.array([0.5, -0.5, -0.0], F))
    assert y[0] == F(0.25) and y[1] == F(-0.25) and bits(y)[2] == 0x80000000
    # SoftClip default mode, level default 0 => bypass (distort.rs:46-50)
    assert np.array_equal(O.Node(O.DISTORT).process(xs), xs)


def test_kat7_fuzz():
    rng = np.random.default_rng(7)
    x = rng.uniform(-1, 1, 128).astype(F)
    y = O.Node(O.DISTORT, [3.0], mode=O.FUZZ).process(x)
    m = M.distort(x, 3.0, O.FUZZ)
    assert np.allclose(y, m, rtol=2e-6, atol=1e-7)
    assert np.all(y <= 0) and abs(np.abs(y).max() - np.abs(x).max()) < 1e-6
    # silent block => NaN (0/0)
    assert np.all(np.isnan(O.Node(O.DISTORT, [3.0], mode=O.FUZZ).process(np.zeros(128, F))))
    # Fuzz ignores the bypass threshold (no `level < 0.001` test in distort.rs:146-172)
    assert not np.array_equal(O.Node(O.DISTORT, [0.0], mode=O.FUZZ).process(x), x)


def test_kat7_overdrive_chebyshev():
    rng = np.random.default_rng(8)
    x = rng.uniform(-1, 1, 128).astype(F)
    y = O.Node(O.OVERDRIVE, [5.0, 0.7, 0.9]).process(x)
    assert np.allclose(y, M.overdrive(x, 5.0, 0.7, 0.9), rtol=3e-7, atol=1e-7)
    assert np.array_equal(O.Node(O.OVERDRIVE, [5.0, 0.7, 0.0009]).process(x), x)
    assert np.array_equal(O.Node(O.OVERDRIVE).process(x), x)  # defaults: level 0 => bypass
    y = O.Node(O.CHEBYSHEV, [4.0, 0.0]).process(x)
    m = M.chebyshev(x, 4.0, 0.0)
    assert np.allclose(y, m, rtol=5e-7, atol=1e-7)
    assert np.array_equal(y[x < 0], x[x < 0])   # level_neg < 0.001 => negative half untouched


def test_kat8_fir():
    rng = np.random.default_rng(9)
    x = rng.uniform(-1, 1, 256).astype(F)
    # default taps [1.0] => identity (fir.rs:61-62)
    assert np.array_equal(O.chain_run([O.Node(O.FIR)], x, link_flags=0), x)
    # integer taps/samples: exact; taps stored reversed (fir.rs:163,168)
    h = np.array([1, 2, 3, 4], np.float64)          # impulse response h[0..3]
    xi = rng.integers(-8, 8, 256).astype(F)
    y = O.chain_run([O.Node(O.FIR, taps_reversed=h[::-1])], xi, link_flags=0)
    causal = np.convolve(xi.astype(np.float64), h)[:256]
    assert np.array_equal(y[3:], causal[3:].astype(F))            # steady state
    # warm-up quirk: state[k] pairs with taps_rev[k]  => y[n] = sum_{k<=n} x[k]*h[T-1-k]
    T = 4
    for nn in range(T - 1):
        e = sum(float(xi[k]) * h[T - 1 - k] for k in range(nn + 1))
        assert y[nn] == F(e)
    assert np.array_equal(y, M.Fir(h[::-1]).run(xi))
    # Average mode: * (1/len as f32) (fir.rs:188)
    ya = O.chain_run([O.Node(O.FIR, mode=O.FIR_AVERAGE, taps_reversed=h[::-1])], xi, link_flags=0)
    assert np.array_equal(ya, (y * (F(1) / F(4))).astype(F))
    # T = 512 random vs exact (fsum) reference: RMS tolerance
    T = 512
    hh = rng.uniform(-1, 1, T) * np.exp(-6.9 * np.arange(T) / T)
    xx = rng.uniform(-1, 1, 1024).astype(F)
    y = O.chain_run([O.Node(O.FIR, taps_reversed=hh[::-1])], xx, link_flags=0)
    ref = np.array([math.fsum(hh[j] * float(xx[n - j]) for j in range(T)) for n in range(T - 1, 1024)])
    err = y[T - 1:].astype(np.float64) - ref
    assert np.sqrt(np.mean(err ** 2)) / np.sqrt(np.mean(ref ** 2)) < 1e-7


def test_kat9_control_port():
    # derive lib.rs:141-148
    ctl = np.array([-1, 0, 1, 2, -3], F)
    n = O.Node(O.GAIN, [1.0])
    y = n.process(np.ones(5, F), ctl=[ctl])
    assert np.array_equal(y, np.array([0, 5, 10, 10, 0], F))
    assert np.array_equal(M.slider_input(ctl, 0, 10), y)
    # latched element 0 is used once the port is disconnected again
    y2 = n.process(np.ones(3, F))
    assert np.array_equal(y2, np.zeros(3, F))
    n = O.Node(O.MIX, [0.5])
    y = n.process(np.full(2, 2, F), np.full(2, 4, F), ctl=[np.array([1, -1], F)])
    assert np.array_equal(y, np.array([4, 2], F))


def test_add_mix():
    rng = np.random.default_rng(11)
    a, b = rng.uniform(-1, 1, 128).astype(F), rng.uniform(-1, 1, 128).astype(F)
    assert np.array_equal(O.Node(O.ADD).process(a, b), M.add(a, b))
    assert np.array_equal(O.Node(O.MIX, [0.3]).process(a, b), M.mix(a, b, 0.3))
    # unconnected "b" => zeros (node.rs:288)
    assert np.array_equal(O.Node(O.ADD).process(a), a)


def test_noise_generator():
    x = O.noise(0x5EED0001, np.arange(4), np.arange(8))
    assert x.shape == (8, 4) and x.dtype == F
    for c in range(4):
        for n in range(8):
            assert x[n, c] == F(O.lib().orc_noise(0x5EED0001, c, n))
    big = O.noise(0x5EED0001, np.arange(256), np.arange(4096))
    assert -1 <= big.min() and big.max() < 1 and abs(big.mean()) < 5e-3
    assert abs(big.var() - 1 / 3) < 5e-3


def test_kat10_signal_gen():
    # signal_gen.rs:57-129.  12 kHz => step 0.25 exactly, so every phase is exact in f32.
    tri = O.Node(O.SIGNAL_GEN, [1.0, 12000.0], mode=O.SIG_TRIANGLE).process(np.zeros(8, F))
    assert np.array_equal(tri, np.array([-0.5, 0, 0.5, -1, -0.5, 0, 0.5, -1], F))   # 2*((k/4)%1)-1
    sq = O.Node(O.SIGNAL_GEN, [0.5, 12000.0], mode=O.SIG_SQUARE).process(np.zeros(8, F))
    # Square compares the block-local `total` (not the phase) with 0.5: low for 2 samples, then high for the rest
    assert np.array_equal(sq, np.array([-0.5, -0.5, 0.5, 0.5, 0.5, 0.5, 0.5, 0.5], F))
    # Constant copies the amplitude block and ignores the input
    assert np.array_equal(O.Node(O.SIGNAL_GEN, [-0.25, 440.0], mode=O.SIG_CONSTANT).process(np.ones(5, F)), np.full(5, -0.25, F))
    # the defaults (signal_gen.rs:41-55): amplitude .5, 100 Hz, Sine
    n = O.Node(O.SIGNAL_GEN)
    y = np.concatenate([n.process(np.zeros(128, F)) for _ in range(8)])
    k = np.arange(1, 1025, dtype=np.float64)
    assert np.abs(y - 0.5 * np.sin(2 * np.pi * 100.0 * k / 48000.0)).max() < 1e-5
    # the clock carries the phase between blocks, wrapped to [0,1): block 2 of a triangle continues block 1
    g, m = O.Node(O.SIGNAL_GEN, [1.0, 1000.0], mode=O.SIG_TRIANGLE), M.SignalGen(1.0, 1000.0, 1)
    for _ in range(6):
        assert np.array_equal(g.process(np.zeros(128, F)), m.process())
    assert 0.0 <= float(m.clock) < 1.0
    # numpy model == C oracle for every mode, with modulated amplitude and frequency (control ports)
    rng = np.random.default_rng(21)
    for mode in range(4):
        g, m = O.Node(O.SIGNAL_GEN, [0.7, 3000.0], mode=mode), M.SignalGen(0.7, 3000.0, mode)
        for blk in range(4):
            a_ctl, f_ctl = rng.uniform(-1.2, 1.2, 128).astype(F), rng.uniform(-1.2, 1.2, 128).astype(F)
            use = blk % 2 == 1
            got = g.process(np.zeros(128, F), ctl=[a_ctl, f_ctl] if use else None)
            if use:
                m.amplitude, m.frequency = M.slider_input(a_ctl, -1, 1)[0], M.slider_input(f_ctl, 0.1, 20000.0)[0]
                want = m.process(128, M.slider_input(a_ctl, -1, 1), M.slider_input(f_ctl, 0.1, 20000.0))
            else:
                want = m.process()
            d = np.abs(bits(got).astype(np.int64) - bits(want).astype(np.int64)).max() if mode == 0 else (0 if np.array_equal(got, want) else 99)
            assert d <= 2, (mode, blk, d)       # glibc sinf is within 1 ulp of the correctly rounded value; x amplitude


def test_kat11_envelope():
    # envelope.rs:34-52 over dasp_envelope 0.11.0 (restated as recalled, oracle/dspfx_oracle.h)
    x = np.array([0.5, -0.25, 0.0, -1.0, 0.125, -0.0], F)
    # defaults attack = release = 0 frames => both gains 0 => env = d + (l - d) * 0 = |x|
    y = O.Node(O.ENVELOPE).process(x)
    assert np.array_equal(y, np.abs(x))
    # attack 0, release n: rises instantly, decays by e^(-1/n) per frame towards |x|
    n = 4.0
    g = F(np.float64(F(2.718281828459045)) ** np.float64(F(-1.0) / F(n)))
    y = O.Node(O.ENVELOPE, [0.0, n]).process(np.array([1, 0, 0, 0, 0.5, 0], F))
    want = [F(1.0)]
    for d in (F(0), F(0), F(0)):
        want.append(F(d + F(F(want[-1] + (-d)) * g)))
    want.append(F(F(0.5) + F(F(want[-1] + F(-0.5)) * (g if not want[-1] < F(0.5) else F(0)))))
    want.append(F(F(0) + F(F(want[-1] + F(-0.0)) * g)))
    assert np.abs(bits(y).astype(np.int64) - bits(np.array(want, F)).astype(np.int64)).max() <= 1   # powf vs f64 pow
    assert abs(float(y[3]) - np.exp(-3 / 4)) < 1e-6
    # attack: approaches a step with time constant n frames; state carries across blocks; reset clears it
    node, m = O.Node(O.ENVELOPE, [10.0, 100.0]), M.Envelope(10.0, 100.0)
    rng = np.random.default_rng(5)
    for _ in range(4):
        blk = rng.uniform(-1, 1, 128).astype(F)
        a, b = node.process(blk), m.process(blk)
        assert np.abs(bits(a).astype(np.int64) - bits(b).astype(np.int64)).max() <= 4    # 1-ulp gain difference, accumulated
    step = O.Node(O.ENVELOPE, [10.0, 0.0]).process(np.ones(64, F))
    k = np.arange(1, 65)
    assert np.abs(step - (1 - np.exp(-k / 10.0))).max() < 1e-5
    node.reset()
    assert np.array_equal(node.process(np.zeros(4, F)), np.zeros(4, F))


def test_kat11_envelope_closed_forms_in_both_restatements():
    """dasp_envelope 0.11.0 Detector::next as written out in oracle/dspfx_oracle.h: a unit step from rest gives
    env[n] = 1 - g_a^(n+1), the release after it env[n] = env0 * g_r^(n+1) -- the C restatement and the numpy one, each
    against the closed form (evaluated in f64 from the f32 gain both of them use)."""
    for attack, release in ((10.0, 100.0), (1.0, 3.0), (250.0, 1000.0)):
        ga = np.float64(M.Envelope.calc_gain(attack))
        gr = np.float64(M.Envelope.calc_gain(release))
        assert abs(ga - math.exp(-1.0 / attack)) < 1e-7 and abs(gr - math.exp(-1.0 / release)) < 1e-7
        x = np.concatenate([np.ones(96, F), np.zeros(96, F)])
        n = np.arange(96)
        rise = 1.0 - ga ** (n + 1)
        fall = rise[-1] * gr ** (n + 1)
        want = np.concatenate([rise, fall])
        for name, got in (("c", np.concatenate([O.Node(O.ENVELOPE, [attack, release]).process(x[:128]), np.zeros(0, F)])),
                          ("numpy", M.Envelope(attack, release).process(x[:128]))):
            assert np.abs(got.astype(np.float64) - want[:128]).max() < 128 * 2.0 ** -23, (name, attack, release)
        # two blocks: the state carries over (envelope.rs keeps the Detector in the node)
        node, model = O.Node(O.ENVELOPE, [attack, release]), M.Envelope(attack, release)
        a = np.concatenate([node.process(x[:128]), node.process(x[128:])])
        b = np.concatenate([model.process(x[:128]), model.process(x[128:])])
        assert np.abs(a.astype(np.float64) - want).max() < 192 * 2.0 ** -23
        assert np.abs(b.astype(np.float64) - want).max() < 192 * 2.0 ** -23


def _sinc_formula(x, ratio, n_out, depth=8):
    """Direct evaluation of dasp's Sinc::interpolate as written out in oracle/dspfx_oracle.h, with no ring and no index
    bookkeeping: output m sits at source time t = m * ratio - depth = j0 + phi; left taps w(phi + n) x[j0 - n], right
    taps w(1 - phi + n) x[j0 + 1 + n], n = 0..7, w(a) = sinc(a) (0.5 + 0.5 cos(pi a / depth)) -- except that the
    outermost right tap (n = 7) would be the frame one past the newest in the 16-frame ring, and the ring indexes modulo
    its length: it reads the OLDEST frame, x[j0 - 8] (weight <= 4e-4).  Frames outside the signal are zeros."""
    def w(a):
        return np.sinc(a) * (0.5 + 0.5 * math.cos(math.pi * a / depth))

    def at(j):
        return x[j] if 0 <= j < len(x) else 0.0
    out = np.zeros(n_out)
    for m in range(n_out):
        t = m * ratio - depth
        j0 = int(math.floor(t))
        phi = t - j0
        acc = 0.0
        for n in range(depth):
            acc += w(phi + n) * at(j0 - n)
            acc += w(1.0 - phi + n) * at(j0 + 1 + n if n < depth - 1 else j0 - depth)
        out[m] = acc
    return out


def test_kat12_sinc_converter_closed_forms():
    """dasp 0.11.0 Converter + Sinc (fir.rs:153-165) as restated in dsp-stuff_amd/ir.py (include/dspfx_ir.hpp is held
    bit-identical to it by tests/test_cpp_ir.py).  (i) At integer rate ratios every output lands on a source frame:
    EXACT, delayed by the ring's depth of 8 frames.  (ii) 44.1 -> 48 kHz: a band-limited sinusoid is interpolated to
    within 2.5e-3 of sin(2 pi f t) (the 16-tap Hann window's ripple; measured 1.2e-3).  (iii) The output equals the direct
    evaluation of the published formula, once the ring has filled."""
    from __graft_entry__ import load_package
    load_package()
    from dsp_stuff_amd import ir
    n = 1500
    k = np.arange(n)
    for f in (100.0, 1000.0, 5000.0, 10000.0):
        # (i) 1:1 and 2:1
        for fs in (48000.0, 96000.0):
            x = np.sin(2 * np.pi * f * k / fs)
            y = ir.resample_dasp_sinc(x, fs, 48000.0)
            t = np.arange(len(y)) * (fs / 48000.0) - 8
            ok = (t > 16) & (t < n - 16)
            assert np.abs(y[ok] - np.sin(2 * np.pi * f * t[ok] / fs)).max() < 1e-12, (f, fs)
        # (ii) 44.1 kHz -> 48 kHz
        x = np.sin(2 * np.pi * f * k / 44100.0)
        y = ir.resample_dasp_sinc(x, 44100.0, 48000.0)
        t = np.arange(len(y)) * (44100.0 / 48000.0) - 8
        ok = (t > 16) & (t < n - 16)
        assert np.abs(y[ok] - np.sin(2 * np.pi * f * t[ok] / 44100.0)).max() < 2.5e-3, f
    # (iii) against the formula itself, on noise (any signal), both rate directions
    rng = np.random.default_rng(12)
    x = rng.uniform(-1, 1, 400)
    for src_hz in (44100.0, 32000.0, 88200.0, 96000.0):
        y = ir.resample_dasp_sinc(x, src_hz, 48000.0)
        want = _sinc_formula(x, src_hz / 48000.0, len(y))
        lo = int(math.ceil(24 * 48000.0 / src_hz))          # the ring is full and the index has reached its depth
        hi = len(y) - lo
        assert np.abs(y[lo:hi] - want[lo:hi]).max() < 1e-12, src_hz


def _both(kind, params, mode=0, delay_len=None, restored=False):
    """The same node in the C restatement and in the numpy one (set_param / process have the same shape in both)."""
    c = O.Node(kind, list(params), mode, delay_len, restored=restored)
    m = M.NodeModel(kind, list(params), mode, delay_len)
    if restored and kind == M.K_REVERB:
        m.impl = M.Reverb(M.delay_len_from_seconds(m.p[1], bool(mode & 1)), m.p[0])
    return c, m


def test_kat13_reverb_any_slider_store_swaps_in_a_zero_ring():
    """reverb.rs:19 attaches Reverb::refresh_seconds to the NODE: the generated render() runs it when ANY widget changed
    (dsp-stuff-derive/src/lib.rs:487-497, 560-568), so a `decay` store replaces the ring by a new zero-filled one
    (reverb.rs:55-71) -- the echo tail is cut.  Closed form: an impulse into a D-sample delay with decay g gives g^k at n = kD;
    a store at n = 1.5 D leaves NOTHING at 2 D, and the node behaves like a fresh one from the store on."""
    D, g = 256, F(0.5)
    x = np.zeros(4 * D, F)
    x[0] = 1.0
    x[2 * D + 5] = 0.25                                              # a second impulse after the store
    for node in _both(O.REVERB, [g], delay_len=D):
        y = np.concatenate([node.process(x[k:k + 128]) for k in range(0, 3 * D // 2, 128)])
        assert y[0] == 1.0 and y[D] == g and np.count_nonzero(y) == 2
        node.set_param(0, 0.25)                                      # the decay slider moves: refresh_seconds
        z = np.concatenate([node.process(x[k:k + 128]) for k in range(3 * D // 2, 4 * D, 128)])
        assert z[D // 2] == 0.0                                      # n = 2 D: the old ring held g^2 here -- gone
        want = np.zeros_like(z)
        want[D // 2 + 5] = 0.25                                      # the new impulse and its echo, with the NEW decay
        want[D // 2 + 5 + D] = F(0.25) * F(0.25)
        assert np.array_equal(bits(z), bits(want))
    # without the hook the tail would be there: the restatement's own process() keeps it
    keep = O.Node(O.REVERB, [g], delay_len=D)
    t = np.concatenate([keep.process(x[k:k + 128]) for k in range(0, 4 * D, 128)])
    assert t[2 * D] == g * g


def test_kat13_reverb_fresh_node_becomes_a_half_second_delay_at_its_first_slider_change():
    """NodeStatic::new leaves make_buffer()'s 128-sample ring (reverb.rs:44-52) under a slider that shows 0.5 s: the first
    widget change runs refresh_seconds with seconds = 0.5 -> max((0.5 * 48000) as usize, 128) = 24000 samples (reverb.rs:58);
    restore() runs the hook once itself (lib.rs:319-337)."""
    x = np.zeros(128, F)
    x[3] = 1.0

    def echo_at(node, blocks):
        y = np.concatenate([node.process(x if k == 0 else np.zeros(128, F)) for k in range(blocks)])
        return np.flatnonzero(y)

    for node in _both(O.REVERB, [0.5, 0.5]):                         # fresh from the menu
        assert list(echo_at(node, 3)) == [3, 131, 259]               # a 128-sample delay
        node.set_param(0, 0.5)                                       # "changed" even to the same value: a new ring, 24000 long
        assert list(echo_at(node, 190)) == [3, 24003]
    for node in _both(O.REVERB, [0.5, 0.5], mode=1):                 # fresh from the menu under the page-rounded reading: make_buffer()
        assert list(echo_at(node, 17)) == [3, 1027, 2051]            # is refresh_seconds' three rivulet calls with 128 -> 1024 samples
        node.set_param(0, 0.5)
        assert list(echo_at(node, 193)) == [3, 24579]                # ... and the half second is 24576
    for node in _both(O.REVERB, [0.5, 0.01], restored=True):         # restored with seconds = 0.01: 480 samples
        assert list(echo_at(node, 5)) == [3, 483]
    for node in _both(O.REVERB, [0.5, 0.01], mode=1, restored=True):  # the page-rounded reading: 1024
        assert list(echo_at(node, 9)) == [3, 1027]
    for node in _both(O.REVERB, [0.5, 0.002], delay_len=300):        # explicit ring, seconds slider 0.002 -> 128 at a store
        assert list(echo_at(node, 4))[:2] == [3, 303]
        node.set_param(1, 0.002)                                     # the seconds slider itself
        assert list(echo_at(node, 3)) == [3, 131, 259]
    for node in _both(O.REVERB, [0.5], delay_len=300):               # no seconds slider with the node: decay stores keep the 300 ...
        node.set_param(0, 0.5)
        assert list(echo_at(node, 4))[:2] == [3, 303]
        node.set_param(1, 0.0)                                       # ... a STORED 0.0 is a value like any other: max(0, 128) = 128 (reverb.rs:58)
        assert list(echo_at(node, 3)) == [3, 131, 259]
        node.set_param(0, 0.5)                                       # and it stays the slider's value for later hooks
        assert list(echo_at(node, 3)) == [3, 131, 259]
    assert M.delay_len_from_seconds(0.5) == O.delay_len(0.5) == 24000 and M.delay_len_from_seconds(0.5, True) == O.delay_len(0.5, True) == 24576
    for s in (0.0, 0.001, 0.0026666, 0.0026667, 0.3333, 1.0):
        assert M.delay_len_from_seconds(s) == O.delay_len(s), s


def test_kat13_biquad_store_resets_state_and_other_nodes_keep_theirs():
    """The other hook (biquad.rs:15, 62-76) and the absence of one: a one-pole's `z` survives a ratio store (low_pass.rs has
    no after_settings_change), a biquad's history does not survive a coefficient store."""
    rng = np.random.default_rng(3)
    x = rng.uniform(-1, 1, 256).astype(F)
    for node in _both(O.BIQUAD, [1.0, -0.5, 0.2, 0.3, 0.2, 0.1]):
        node.process(x[:128])
        node.set_param(3, 0.3)                                       # same value: still "changed" -> reset_state
        fresh = O.Node(O.BIQUAD, [1.0, -0.5, 0.2, 0.3, 0.2, 0.1])
        assert np.array_equal(bits(node.process(x[128:])), bits(fresh.process(x[128:])))
    for node in _both(O.LOW_PASS, [0.9]):
        node.process(x[:128])
        node.set_param(0, 0.5)
        fresh = O.Node(O.LOW_PASS, [0.5])
        assert not np.array_equal(bits(node.process(x[128:])), bits(fresh.process(x[128:])))
