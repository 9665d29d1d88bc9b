"""GPU parity: the HIP path (through the C ABI) against the CPU oracle on the same
seeded inputs.  Bars (BASELINE.json north_star): bit-exact for gain / pass-through /
indexing; <= 1 ulp f32 for IIR / delay / arithmetic waveshapers; <= LIBM_ULP for the
modes that call libm (glibc on the reference side, ocml on the GPU); stated RMS
tolerance for FIR."""
import numpy as np
import pytest

import oracle as O
from chains import chain3, chain5, fir_taps, ulp_diff

pytestmark = pytest.mark.gpu
F = np.float32
# libm-backed modes: GPU = f64 evaluation rounded once (correctly rounded f32); glibc's f32
# routines are within 1 ulp (sinf/atanf/expf) resp. 2 ulp (tanhf) of that (measured)
LIBM_ULP = {2: 2, 5: 1, 6: 1}   # distort mode -> bar
LIBM_COMPOSITE_ULP = 4          # overdrive / chebyshev node: libm result feeds further f32 ops
FIR_RMS_TOL = 1e-6  # relative RMS error of the f32 MFMA path vs the f64-accumulating oracle (measured 3.2e-7 at T=4096)


@pytest.fixture(scope="module")
def torch_cuda():
    import torch
    assert torch.cuda.is_available(), "gpu tests need a GPU"
    return torch


def run_gpu(dspfx, torch, chain, x, link_flags=3, block=128, side=None, want_mix=False, max_frames=None, tile=0):
    """x: [frames][N] numpy -> GPU engine block by block (device path).  `tile` selects the
    channel-tiled HBM layout [N/W][block][W]; every block is handed over in that layout."""
    nf, N = x.shape
    eng = dspfx.Engine(N, max_frames or block, link_flags=link_flags, tile_channels=tile)
    eng.set_chain(chain)
    y = np.empty_like(x)
    mix = np.empty(nf, np.float32)
    for f0 in range(0, nf, block):
        n = min(block, nf - f0)
        dx = torch.from_numpy(dspfx.to_layout(x[f0:f0 + n], tile)).cuda()
        ds = torch.from_numpy(dspfx.to_layout(side[f0:f0 + n], tile)).cuda() if side is not None else None
        dy = torch.empty_like(dx)
        dm = torch.empty(n, dtype=torch.float32, device="cuda") if want_mix else None
        eng.process(dx, out=dy, side=ds, mix=dm, n_frames=n)
        torch.cuda.synchronize()
        y[f0:f0 + n] = dspfx.from_layout(dy.cpu().numpy(), n, N, tile)
        if want_mix:
            mix[f0:f0 + n] = dm.cpu().numpy()
    eng.close()
    return (y, mix) if want_mix else y


def run_oracle(chain, x, link_flags=3, side=None):
    return O.run_channels([n.oracle_desc() for n in chain], x, link_flags, side)


def noise_block(N, nf, seed=0x5EED0001, c0=0, n0=0):
    return O.noise(seed, np.arange(c0, c0 + N), np.arange(n0, n0 + nf))


def test_golden_vectors(dspfx, torch_cuda):
    """The HIP path against the committed golden vectors (tests/golden/), one channel replicated to 64."""
    from golden_util import bar_for, load_all
    for g in load_all():
        chain = [dspfx.NodeSpec(d["kind"], d.get("params") or [], d.get("mode") or 0, d.get("delay_len") or 0,
                                d.get("taps_reversed")) for d in g["descs"]]
        x = np.repeat(g["x"][:, None], 64, axis=1)
        y = run_gpu(dspfx, torch_cuda, chain, x, g["link_flags"])
        assert np.array_equal(y, np.repeat(y[:, :1], 64, axis=1)), g["name"]      # channels are independent
        bar = bar_for(g["name"])
        if bar is None:
            assert np.abs(y[:, 0] - g["y"]).max() <= 4e-6 * np.abs(g["y"]).max(), g["name"]
        else:
            assert ulp_diff(y[:, 0], g["y"]).max() <= bar, (g["name"], ulp_diff(y[:, 0], g["y"]).max())


# ---------------------------------------------------------------- single nodes

def test_gain_bit_exact(dspfx, torch_cuda):
    x = noise_block(256, 256)
    x[0, :8] = [0.0, -0.0, 1e-42, -1e-42, np.inf, -np.inf, 3.4e38, 1e-38]
    for level in (0.0, 0.8, 1.0, 10.0):
        y = run_gpu(dspfx, torch_cuda, [dspfx.Gain(level)], x, link_flags=0)
        with np.errstate(invalid="ignore"):
            assert np.array_equal(y.view(np.uint32), (x * F(level)).view(np.uint32)) or \
                ulp_diff(y, x * F(level)).max() == 0


def test_passthrough_and_link_scale(dspfx, torch_cuda):
    x = noise_block(192, 128)
    x[0, 0], x[0, 1] = 0.5, -0.0
    y = run_gpu(dspfx, torch_cuda, [dspfx.Gain(1.0)], x, link_flags=0)
    assert np.array_equal(y.view(np.uint32), x.view(np.uint32))              # bit-exact pass-through
    y = run_gpu(dspfx, torch_cuda, [dspfx.Gain(1.0)], x, link_flags=2)
    assert float(y[0, 0]).hex() == "0x1.fff2e40000000p-2"                   # KAT-2
    assert y.view(np.uint32)[0, 1] == 0                                      # 0.0 + -0.0 = +0.0
    ref = run_oracle([dspfx.Gain(1.0)], x, 2)
    assert np.array_equal(y.view(np.uint32), ref.view(np.uint32))
    # empty chain == wire
    y = run_gpu(dspfx, torch_cuda, [], x, link_flags=3)
    assert np.array_equal(y.view(np.uint32), x.view(np.uint32))


@pytest.mark.parametrize("N", [1, 63, 64, 65, 200, 256, 1000])
def test_ragged_channel_counts(dspfx, torch_cuda, N):
    """N not a multiple of the wave width exercises the guarded tail launch."""
    x = noise_block(N, 256)
    ch = chain5(dspfx, delay=128)
    y, mix = run_gpu(dspfx, torch_cuda, ch, x, want_mix=True)
    ref = run_oracle(ch, x)
    assert ulp_diff(y, ref).max() <= 1
    assert np.allclose(mix, ref.astype(np.float64).sum(axis=1), rtol=1e-5, atol=1e-4)


def test_biquad_defaults_and_reset(dspfx, torch_cuda):
    x = np.zeros((128, 64), F)
    x[0] = 1
    y = run_gpu(dspfx, torch_cuda, [dspfx.BiQuad()], x, link_flags=0)
    ref = run_oracle([dspfx.BiQuad()], x, 0)
    assert ulp_diff(y, ref).max() <= 1
    assert abs(float(y[5, 0]) - 0.758 * 0.24 ** 5) < 1e-7                   # KAT-3
    # set_param zeroes the state (biquad.rs:74)
    eng = dspfx.Engine(64, 128, link_flags=0)
    eng.set_chain([dspfx.BiQuad()])
    dx = torch_cuda.from_numpy(x).cuda()
    dy = torch_cuda.empty_like(dx)
    eng.process(dx, out=dy)
    eng.set_param(0, 3, 0.758)
    eng.process(dx, out=dy)
    torch_cuda.cuda.synchronize()
    assert np.array_equal(dy.cpu().numpy(), y)


def test_biquad_random_coeffs(dspfx, torch_cuda):
    rng = np.random.default_rng(4)
    x = noise_block(128, 512)
    for _ in range(4):
        r, th, a0 = rng.uniform(0.1, 0.95), rng.uniform(0.1, 3.0), rng.uniform(0.5, 2.0)
        p = [a0, -2 * r * np.cos(th) * a0, r * r * a0] + list(rng.uniform(-1, 1, 3) * a0)
        ch = [dspfx.BiQuad(*p)]
        assert ulp_diff(run_gpu(dspfx, torch_cuda, ch, x, 0), run_oracle(ch, x, 0)).max() <= 1


def test_one_pole(dspfx, torch_cuda):
    x = noise_block(128, 384)
    for r in (0.0, 0.5, 0.93, 1.0):
        for mk in (dspfx.LowPass, dspfx.HighPass):
            ch = [mk(r)]
            assert ulp_diff(run_gpu(dspfx, torch_cuda, ch, x, 0), run_oracle(ch, x, 0)).max() <= 1


@pytest.mark.parametrize("D", [128, 1024, 24000])
def test_delay_impulse(dspfx, torch_cuda, D):
    nblk = (3 * D) // 128 + 2
    x = np.zeros((nblk * 128, 64), F)
    x[0] = 1
    y = run_gpu(dspfx, torch_cuda, [dspfx.Reverb(delay_samples=D, decay=0.5)], x, 0)
    e = np.zeros_like(x)
    for k in range(len(e) // D + 1):
        if k * D < len(e):
            e[k * D] = 0.5 ** k
    assert np.array_equal(y, e)                                              # KAT-6: exact


def test_delay_block_size_invariance(dspfx, torch_cuda):
    """B=64/128/256/384 give identical results, incl. B > D (engine splits at D)."""
    x = noise_block(128, 768)
    ch = [dspfx.Reverb(delay_samples=128, decay=0.7), dspfx.BiQuad()]
    ys = [run_gpu(dspfx, torch_cuda, ch, x, 3, block=b) for b in (64, 128, 256, 384)]
    ref = run_oracle(ch, x, 3)
    for y in ys:
        assert np.array_equal(y, ys[0])
    assert ulp_diff(ys[0], ref).max() <= 1


@pytest.mark.parametrize("mode", [0, 1, 3, 7, 8])
def test_distort_arithmetic_modes(dspfx, torch_cuda, mode):
    x = noise_block(128, 128) * F(2.5)
    x[0, :10] = [0, -0.0, 0.25, -0.25, 0.5, -0.5, 1, -1, 2, -2]
    for L in (0.0, 0.0009, 0.001, 1.0, 3.0, 30.0):
        ch = [dspfx.Distort(L, mode)]
        y, ref = run_gpu(dspfx, torch_cuda, ch, x, 0), run_oracle(ch, x, 0)
        assert ulp_diff(y, ref).max() <= 1, (mode, L)
        if L < 0.001:
            assert np.array_equal(y.view(np.uint32), x.view(np.uint32))      # bypass is bit-exact


@pytest.mark.parametrize("mode", [2, 5, 6])
def test_distort_libm_modes(dspfx, torch_cuda, mode):
    x = noise_block(128, 128)
    for L in (1.0, 3.0, 30.0):
        ch = [dspfx.Distort(L, mode)]
        y, ref = run_gpu(dspfx, torch_cuda, ch, x, 0), run_oracle(ch, x, 0)
        assert ulp_diff(y, ref).max() <= LIBM_ULP[mode], (mode, L, ulp_diff(y, ref).max())


def test_fuzz(dspfx, torch_cuda):
    x = noise_block(100, 256)
    ch = [dspfx.Gain(0.9), dspfx.Distort(3.0, dspfx.FUZZ), dspfx.Gain(0.5)]
    y, ref = run_gpu(dspfx, torch_cuda, ch, x, 3), run_oracle(ch, x, 3)
    # three block-global normalisations and the cancellation in 1 - exp(-|q|) amplify a
    # 1-ulp expf difference, so the bar is relative to the block's peak, not per-sample ulp
    assert np.abs(y - ref).max() <= 4e-6 * np.abs(ref).max()
    assert (ulp_diff(y, ref) > 1).mean() < 0.02
    # silent block => NaN (0/0), like the reference
    z = np.zeros((128, 64), F)
    y = run_gpu(dspfx, torch_cuda, [dspfx.Distort(3.0, dspfx.FUZZ)], z, 0)
    assert np.all(np.isnan(y))
    # Fuzz needs whole 128-frame reference blocks
    eng = dspfx.Engine(64, 128)
    eng.set_chain([dspfx.Distort(3.0, dspfx.FUZZ)])
    t = torch_cuda.zeros((64, 64), device="cuda")
    with pytest.raises(dspfx.DspfxError):
        eng.process(t, n_frames=64)


def test_overdrive_chebyshev(dspfx, torch_cuda):
    x = noise_block(128, 128)
    for ch in ([dspfx.Overdrive(5.0, 0.7, 0.9)], [dspfx.Overdrive(5.0, 0.7, 0.0009)], [dspfx.Overdrive()],
               [dspfx.Chebyshev(4.0, 0.0)], [dspfx.Chebyshev(2.0, 7.5)], [dspfx.Chebyshev()]):
        y, ref = run_gpu(dspfx, torch_cuda, ch, x, 0), run_oracle(ch, x, 0)
        assert ulp_diff(y, ref).max() <= LIBM_COMPOSITE_ULP, ch


def test_add_mix_side_input(dspfx, torch_cuda):
    x, s = noise_block(128, 256), noise_block(128, 256, seed=77)
    for ch in ([dspfx.Add()], [dspfx.Gain(0.5), dspfx.Mix(0.3), dspfx.BiQuad()]):
        for lf in (0, 3):
            y, ref = run_gpu(dspfx, torch_cuda, ch, x, lf, side=s), run_oracle(ch, x, lf, side=s)
            assert ulp_diff(y, ref).max() <= 1
        y, ref = run_gpu(dspfx, torch_cuda, ch, x, 3), run_oracle(ch, x, 3)      # unconnected "b" => zeros
        assert ulp_diff(y, ref).max() <= 1


def test_fast_constant_division_is_exhaustively_exact(dspfx, torch_cuda, monkeypatch):
    """The (double)x * (1/c) division is proven per constant over all 2^32 inputs; constants
    that have exact-tie quotients in the subnormal range fail the proof and keep IEEE division."""
    assert dspfx.verify_fast_division(float(dspfx.link_divisor(1))) == 0      # 1.0001f: every chain hop
    assert dspfx.verify_fast_division(3.0) == 0                               # SoftClip's powi(3)/3.0
    for level in (0.5, 1.0, 2.5, 3.0, 30.0, 0.001):
        assert dspfx.verify_fast_division(level) == 0, level
    # constants whose quotients can be exact ties in the subnormal range (even integers) may fail
    # the proof, depending on which way RN_f64(1/c) errs; any that do must take the IEEE path
    failing = [c for c in (6.0, 10.0, 12.0, 14.0, 18.0, 20.0, 22.0, 24.0, 26.0, 28.0, 30.0)
               if dspfx.verify_fast_division(c) > 0]
    print("constants failing the exhaustive proof:", failing)
    x = noise_block(128, 256)
    x[0, :6] = [4.2e-45, -4.2e-45, 1e-44, 3e-39, 1.5e-44, 2.9e-44]            # subnormal inputs incl. tie cases
    for level in failing[:2] + [6.0, 3.0]:
        for mode in (dspfx.HARD_CLIP, dspfx.SOFT_CLIP):
            ch = [dspfx.Distort(level, mode), dspfx.Gain(1.0)]
            y, ref = run_gpu(dspfx, torch_cuda, ch, x, 3), run_oracle(ch, x, 3)
            assert ulp_diff(y, ref).max() == 0, (level, mode)
    # the forced-IEEE form of the same chain gives the same bits as the fast one: DSPFX_FAST_DIV is read when a chain
    # is planned, so an engine created under it runs the interpreter with every division in its IEEE form
    ch = chain5(dspfx, delay=128)
    y_fast = run_gpu(dspfx, torch_cuda, ch, x, 3)
    eng = dspfx.Engine(x.shape[1], 128)
    eng.set_chain(ch)
    fast_plan = eng.describe()
    monkeypatch.setenv("DSPFX_FAST_DIV", "0")
    eng2 = dspfx.Engine(x.shape[1], 128)
    eng2.set_chain(ch)
    assert "dyn" in eng2.describe() and eng2.describe() != fast_plan       # really another kernel: the IEEE interpreter
    y_ieee = run_gpu(dspfx, torch_cuda, ch, x, 3)
    monkeypatch.delenv("DSPFX_FAST_DIV")
    assert np.array_equal(y_fast.view(np.uint32), y_ieee.view(np.uint32))
    assert ulp_diff(y_fast, run_oracle(ch, x, 3)).max() <= 1


@pytest.mark.jit_frozen          # (it compares dspfx_describe before and after the stores: the kernel must not be swapped meanwhile)
def test_distort_level_slider_moves_without_replanning(dspfx, torch_cuda):
    """Whether x / level may take the exact-product form is decided on the host: only even integers can have
    exact ties among their subnormal quotients.  Any level gives the oracle's bits; a slider store keeps the kernel
    unless it moves to / from an even integer that failed the exhaustive check."""
    x = noise_block(128, 128)
    x[0, :6] = [4.2e-45, -4.2e-45, 1e-44, 3e-39, 1.5e-44, 2.9e-44]
    failing = [c for c in (6.0, 10.0, 12.0, 14.0) if dspfx.verify_fast_division(c) > 0]
    for mode in (dspfx.HARD_CLIP, dspfx.SOFT_CLIP):
        ch = [dspfx.Distort(3.0, mode), dspfx.Gain(1.0)]
        eng = dspfx.Engine(128, 128)
        eng.set_chain(ch)
        plan = lambda: [l for l in eng.describe().splitlines() if l.startswith("stage")]   # (the counters of the process-wide
        plan0 = plan()                                # compiler in the last line move while earlier tests' shapes are still compiling)
        dx = torch_cuda.from_numpy(x).cuda()
        for level in (0.37, 30.0 - 2 ** -19, 7.0, 0.0005, 2.5, 1e-3, 4.0, 16.0) + tuple(failing[:1]) + (3.0,):
            eng.set_param(0, 0, level)
            assert (plan() == plan0) == (level not in failing), level
            y = eng.process(dx, out=torch_cuda.empty_like(dx), n_frames=128).cpu().numpy()
            ref = run_oracle([dspfx.Distort(level, mode), dspfx.Gain(1.0)], x, 3)
            assert ulp_diff(y, ref).max() == 0, (mode, level)


def test_only_even_integers_can_fail_the_division_proof(dspfx, torch_cuda):
    """The host rule behind the fast constant division: constants that are not even integers never fail the
    exhaustive 2^32-input check (their subnormal quotients have no exact ties).  Checked on the device for a
    seeded sample of slider values, link divisors and awkward constants."""
    rng = np.random.default_rng(2)
    consts = [float(dspfx.link_divisor(k)) for k in (1, 2, 3, 5, 16)] + [3.0, 5.0, 7.0, 9.0, 15.0, 29.0, 1.5, 2.5, 6.5,
              0.001, 1e-3 + 1e-9, 30.0 - 2 ** -19, 1.0, 2.0, 4.0, 8.0, 16.0, 0.5, 0.25, 3.0e-5, 12345.678, 2.0 ** 24 + 2]
    consts += [float(np.float32(v)) for v in rng.uniform(0.001, 30.0, 12)]
    consts += [float(np.float32(v)) for v in 2.0 ** rng.uniform(-20, 20, 6)]
    for c in consts:
        even_int = c >= 2 and (c / 2) == int(c / 2) and (c / 2) < 2 ** 31
        pow2 = np.frexp(c)[0] == 0.5
        if even_int and not pow2:
            continue
        assert dspfx.verify_fast_division(c) == 0, c


def test_fuzz_level_control_port(dspfx, torch_cuda):
    """The reference maps the level port for EVERY distort mode before the mode switch (distort.rs:176-180) and fuzz
    zips it per sample (154-160): a connected port modulates the level per sample, latches the first value of each
    128-frame block per channel, and the latch keeps applying after the port is disconnected."""
    N, B, blocks = 96, 128, 5
    x = noise_block(N, B * blocks)
    sig = (noise_block(N, B * blocks, seed=31) * F(1.4)).astype(F)          # exceeds [-1, 1]: exercises the clamp
    for lf, tile in ((3, 0), (0, 0), (3, 64)):
        Nn = N if not tile else 128
        xx, ss = (x[:, :Nn], sig[:, :Nn]) if Nn <= N else (noise_block(Nn, B * blocks), (noise_block(Nn, B * blocks, seed=31) * F(1.4)).astype(F))
        chain = [dspfx.Gain(0.9), dspfx.Distort(3.0, dspfx.FUZZ), dspfx.BiQuad()]
        eng = dspfx.Engine(Nn, B, link_flags=lf, tile_channels=tile)
        eng.set_chain(chain)
        descs = [n.oracle_desc() for n in chain]
        nodes = []
        got, ref = np.empty_like(xx), np.empty_like(xx)

        def run(b, connected):
            sl = slice(b * B, (b + 1) * B)
            dx = torch_cuda.from_numpy(dspfx.to_layout(xx[sl], tile)).cuda()
            dc = torch_cuda.from_numpy(dspfx.to_layout(ss[sl], tile)).cuda()
            dy = torch_cuda.empty_like(dx)
            eng.process(dx, out=dy, n_frames=B, ctl={(1, 0): dc} if connected else None)
            torch_cuda.cuda.synchronize()
            got[sl] = dspfx.from_layout(dy.cpu().numpy(), B, Nn, tile)
            ref[sl] = O.run_channels(descs, xx[sl], lf, ctl={(1, 0): ss[sl]} if connected else None, nodes_out=nodes)

        run(0, True)
        run(1, True)
        run(2, False)                      # disconnected: the per-channel latched levels apply
        eng.set_param(1, 0, 2.5)           # a slider store overwrites the latch
        for chn in nodes:
            chn[1].set_param(0, 2.5)
        run(3, False)
        run(4, True)
        # a channel whose latched level is 0 (its control signal was <= -1 at the block's first sample) goes silent
        # inside fuzz in block 2: 0/0 = NaN like the reference, and the biquad behind it keeps the NaN
        nan = np.isnan(ref)
        assert np.array_equal(np.isnan(got), nan) and 0 < nan.mean() < 0.5
        fin = ~nan
        assert np.abs(got[fin] - ref[fin]).max() <= 4e-6 * np.abs(ref[fin]).max(), (lf, tile, np.abs(got[fin] - ref[fin]).max())
        assert (ulp_diff(got[fin], ref[fin]) > 4).mean() < 0.02
        # the modulation really matters: a constant level gives something else
        const = run_oracle(chain, xx[:B], lf)
        assert np.abs(const - ref[:B]).max() > 1e-3


def test_control_ports(dspfx, torch_cuda):
    """`as_input` sliders fed by links (dsp-stuff-derive/src/lib.rs:122-161): per-sample map
    [-1,1] -> slider range, per-channel latch of each block's first value, which keeps applying
    after the port is disconnected until a slider store overwrites it."""
    N, B, blocks = 96, 128, 6
    rng = np.random.default_rng(12)
    x, side = noise_block(N, B * blocks), noise_block(N, B * blocks, seed=5)
    sig = lambda seed: (noise_block(N, B * blocks, seed=seed) * F(1.5)).astype(F)   # exceeds [-1,1]: exercises the clamp
    cg, cd, cb, cdr, cl, cr = (sig(s) for s in (21, 22, 23, 24, 25, 26))
    cl = (cl * F(0.2) + F(0.2)).astype(F)                                            # keep overdrive level mostly > 0.001
    # chain A: arithmetic-only nodes => the <= 1 ulp bar holds with modulated sliders too;
    # chain B adds the libm-backed ones (overdrive's atan, tanh): their 1-ulp differences vs glibc pass
    # through a mix and a biquad, so the bar is absolute (a few 1e-7 at unit scale), not per-sample ulp
    chain_a = [dspfx.Gain(1.0), dspfx.Distort(3.0, dspfx.HARD_CLIP), dspfx.Mix(0.5), dspfx.BiQuad(),
               dspfx.Distort(2.0, dspfx.SOFT_CLIP)]
    ctl_a = {(0, 0): cg, (1, 0): cd, (2, 0): cr, (4, 0): cb}
    chain_b = [dspfx.Gain(1.0), dspfx.Distort(3.0, dspfx.HARD_CLIP), dspfx.Overdrive(5.0, 0.5, 0.8), dspfx.Mix(0.5),
               dspfx.BiQuad(), dspfx.Distort(2.0, dspfx.TANH)]
    ctl_b = {(0, 0): cg, (1, 0): cd, (2, 0): cb, (2, 1): cdr, (2, 2): cl, (3, 0): cr, (5, 0): cd}
    for chain, ctl_all, exact in ((chain_a, ctl_a, True), (chain_b, ctl_b, False)):
        keys = list(ctl_all)
        for lf in (3, 0):
            eng = dspfx.Engine(N, B, link_flags=lf)
            eng.set_chain(chain)
            nodes = []
            descs = [n.oracle_desc() for n in chain]
            dev = {k: torch_cuda.from_numpy(v).cuda() for k, v in ctl_all.items()}
            dx, ds = torch_cuda.from_numpy(x).cuda(), torch_cuda.from_numpy(side).cuda()
            dy = torch_cuda.empty_like(dx)
            ref = np.empty_like(x)

            def run(b, ks):
                sl = slice(b * B, (b + 1) * B)
                eng.process(dx[sl], out=dy[sl], side=ds[sl], n_frames=B, ctl={k: dev[k][sl] for k in ks} or None)
                ref[sl] = O.run_channels(descs, x[sl], lf, side[sl], ctl={k: ctl_all[k][sl] for k in ks} or None,
                                         nodes_out=nodes)

            run(0, keys)                                # every port connected
            run(1, keys)
            run(2, keys[:2])                            # others disconnected: their latched per-channel values apply
            run(3, [])                                  # nothing connected
            eng.set_param(0, 0, 0.7)                    # a slider store overwrites the latch (lib.rs:487-492)
            for chn in nodes:
                chn[0].set_param(0, 0.7)
            run(4, [])
            run(5, keys[-1:])
            torch_cuda.cuda.synchronize()
            got = dy.cpu().numpy()
            if exact:
                d = ulp_diff(got, ref)
                assert d.max() <= 1, (lf, d.max(), np.unravel_index(d.argmax(), d.shape))
            else:
                assert np.abs(got - ref).max() <= 2e-6, (lf, np.abs(got - ref).max())
                assert (ulp_diff(got, ref) > LIBM_COMPOSITE_ULP).mean() < 0.01
    # no side input, channel-tiled layout, a block longer than the delay line (engine splits it at D):
    # the control link still gets its collect_and_average hop and its signal is offset with the samples
    chain = [dspfx.Gain(1.0), dspfx.Reverb(delay_samples=128, decay=0.4), dspfx.Distort(2.0, dspfx.HARD_CLIP)]
    ctl_c = {(0, 0): cg, (2, 0): cd}
    Nc = 64
    xc = np.ascontiguousarray(x[:, :Nc])
    ctl_c = {k: np.ascontiguousarray(v[:, :Nc]) for k, v in ctl_c.items()}
    for lf in (3, 0):
        for tile in (0, 64):
            eng = dspfx.Engine(Nc, 256, link_flags=lf, tile_channels=tile)
            eng.set_chain(chain)
            got = np.empty_like(xc)
            for b in range(0, B * blocks, 256):
                sl = slice(b, b + 256)
                dxx = torch_cuda.from_numpy(dspfx.to_layout(xc[sl], tile)).cuda()
                dc = {k: torch_cuda.from_numpy(dspfx.to_layout(v[sl], tile)).cuda() for k, v in ctl_c.items()}
                eng.process(dxx, n_frames=256, ctl=dc)
                torch_cuda.cuda.synchronize()
                got[sl] = dspfx.from_layout(dxx.cpu().numpy(), 256, Nc, tile)
            ref = O.run_channels([n.oracle_desc() for n in chain], xc, lf, ctl=ctl_c)
            assert ulp_diff(got, ref).max() <= 1, (lf, tile)
    # bad slider indices are rejected (Fuzz's level port itself is a real port: test_fuzz_level_control_port)
    eng = dspfx.Engine(N, B)
    eng.set_chain([dspfx.Distort(3.0, dspfx.FUZZ), dspfx.BiQuad()])
    t = torch_cuda.zeros((B, N), device="cuda")
    with pytest.raises(dspfx.DspfxError):
        eng.process(t, n_frames=B, ctl={(0, 1): t})
    with pytest.raises(dspfx.DspfxError):
        eng.process(t, n_frames=B, ctl={(1, 0): t})


def test_imported_dspconfig_runs_on_gpu(dspfx, torch_cuda):
    """A graph in the reference's save format -> chain -> engine == oracle."""
    from dsp_stuff_amd import config
    from test_config_ir_cpu import REFERENCE_STYLE
    import json
    chain, _ = config.load_dspconfig(json.dumps(REFERENCE_STYLE))
    chain[2] = dspfx.Reverb(delay_samples=256, decay=chain[2].params[0])     # keep the test ring small
    x = noise_block(128, 512)
    y, ref = run_gpu(dspfx, torch_cuda, chain, x, 3), run_oracle(chain, x, 3)
    assert ulp_diff(y, ref).max() <= 1


# ----------------------------------------------------------------------- chains

def test_config1_single_channel_chain(dspfx, torch_cuda):
    """BASELINE config 1: 1 channel, gain -> biquad LP -> delay(24000), 128-frame blocks."""
    ch = chain3(dspfx)
    x = noise_block(1, 128 * 400)            # > 2 delay periods
    y, ref = run_gpu(dspfx, torch_cuda, ch, x, 3), run_oracle(ch, x, 3)
    assert ulp_diff(y, ref).max() <= 1


def test_chain3_and_chain5_all_variants(dspfx, torch_cuda, monkeypatch):
    """Every compiled kernel variant (static/dynamic, F, CPL) gives the same bits."""
    x = noise_block(512, 512)
    for mk in (chain3, chain5):
        ch = mk(dspfx, delay=256)
        for lf in (0, 1, 3):
            ref = run_oracle(ch, x, lf)
            base = None
            for var in ("static=0,f=8,cpl=1", "static=0,f=4,cpl=1", "static=0,f=16,cpl=1", "static=0,f=8,cpl=2", "static=0,f=4,cpl=2",
                        "static=1,f=8,cpl=1", "static=1,f=8,cpl=2", "static=1,f=32,cpl=1",
                        "static=1,f=8,cpl=4", "static=1,f=16,cpl=1", "static=1,f=16,cpl=2", "static=1,f=4,cpl=4"):
                monkeypatch.setenv("DSPFX_VARIANT", var)
                y = run_gpu(dspfx, torch_cuda, ch, x, lf)
                assert ulp_diff(y, ref).max() <= 1, (mk.__name__, lf, var)
                if base is None:
                    base = y
                assert np.array_equal(y.view(np.uint32), base.view(np.uint32)), (mk.__name__, lf, var)
        monkeypatch.delenv("DSPFX_VARIANT", raising=False)


def test_long_chain_splits_into_stages(dspfx, torch_cuda):
    """> 8 nodes => several fused launches; Fuzz in the middle => its own stage."""
    ch = [dspfx.Gain(0.9), dspfx.BiQuad(), dspfx.LowPass(0.3), dspfx.HighPass(0.9), dspfx.Distort(2.0, dspfx.HARD_CLIP),
          dspfx.Reverb(delay_samples=128, decay=0.4), dspfx.Distort(1.5, dspfx.SQUARE), dspfx.BiQuad(1, -0.5, 0.1, 0.3, 0.2, 0.1),
          dspfx.Gain(1.1), dspfx.Reverb(delay_samples=384, decay=0.3), dspfx.LowPass(0.1), dspfx.Distort(4.0, dspfx.RECIP_SOFT_CLIP)]
    x = noise_block(192, 512)
    y, ref = run_gpu(dspfx, torch_cuda, ch, x, 3), run_oracle(ch, x, 3)
    assert ulp_diff(y, ref).max() <= 1


def test_state_export_import_continues(dspfx, torch_cuda):
    """Run k blocks, export state, import into a fresh engine, continue: identical."""
    ch = chain5(dspfx, delay=256)
    x = noise_block(128, 1024)
    full = run_gpu(dspfx, torch_cuda, ch, x, 3)
    a = dspfx.Engine(128, 128)
    a.set_chain(ch)
    dx = torch_cuda.from_numpy(x).cuda()
    dy = torch_cuda.empty_like(dx)
    for f0 in range(0, 512, 128):
        a.process(dx[f0:f0 + 128], out=dy[f0:f0 + 128])
    states = [a.state_export(i) for i in range(len(ch))]
    assert [len(s) for s in states] == [4 * 128 * 4, 0, 256 * 128 * 4, 4 * 128 * 4, 0]
    b = dspfx.Engine(128, 128)
    b.set_chain(ch)
    for i, s in enumerate(states):
        if len(s):
            b.state_import(i, s)
    for f0 in range(512, 1024, 128):
        b.process(dx[f0:f0 + 128], out=dy[f0:f0 + 128])
    torch_cuda.cuda.synchronize()
    assert np.array_equal(dy.cpu().numpy()[512:], full[512:])
    # delay state export is oldest-first: row 0 == output of frame 512-256
    ring = states[2].view(np.float32).reshape(256, 128)
    st_only = run_oracle(ch[:3], x[:512], 3)
    assert ulp_diff(ring, st_only[256:512]).max() <= 1


def test_state_export_is_layout_independent(dspfx, torch_cuda):
    """Exported state is canonical ([D][N] oldest-first for a delay line) whatever the engine's HBM layout,
    so it can move between a frame-major and a channel-tiled engine."""
    N, B = 256, 128
    ch = chain5(dspfx, delay=384)
    x = noise_block(N, B * 8)
    full = run_gpu(dspfx, torch_cuda, ch, x, 3)
    exports = {}
    for tile in (0, 64):
        eng = dspfx.Engine(N, B, tile_channels=tile)
        eng.set_chain(ch)
        for b in range(5):      # 640 frames: ring position is mid-ring (640 % 384 = 256)
            blk = torch_cuda.from_numpy(dspfx.to_layout(x[b * B:(b + 1) * B], tile)).cuda()
            eng.process(blk, n_frames=B)
        torch_cuda.cuda.synchronize()
        exports[tile] = [eng.state_export(i) for i in range(len(ch))]
    for a, b in zip(exports[0], exports[64]):
        assert np.array_equal(a, b)
    # frame-major state into a tiled engine (and back), then continue: same output as the uninterrupted run
    for src, dst_tile in ((0, 64), (64, 0)):
        eng = dspfx.Engine(N, B, tile_channels=dst_tile)
        eng.set_chain(ch)
        for i, st in enumerate(exports[src]):
            if len(st):
                eng.state_import(i, st)
        out = []
        for b in range(5, 8):
            blk = torch_cuda.from_numpy(dspfx.to_layout(x[b * B:(b + 1) * B], dst_tile)).cuda()
            eng.process(blk, n_frames=B)
            torch_cuda.cuda.synchronize()
            out.append(dspfx.from_layout(blk.cpu().numpy(), B, N, dst_tile))
        assert np.array_equal(np.concatenate(out), full[5 * B:])


def test_mix_bus_and_finish(dspfx, torch_cuda):
    N = 4096
    ch = chain5(dspfx, delay=128)
    x = noise_block(N, 256)
    y, mix = run_gpu(dspfx, torch_cuda, ch, x, 3, want_mix=True)
    ref = run_oracle(ch, x, 3)
    assert ulp_diff(y, ref).max() <= 1
    exact = ref.astype(np.float64).sum(axis=1)
    assert np.allclose(mix, exact, rtol=1e-5, atol=1e-3)
    # deterministic: same bits on a second run
    y2, mix2 = run_gpu(dspfx, torch_cuda, ch, x, 3, want_mix=True)
    assert np.array_equal(mix.view(np.uint32), mix2.view(np.uint32))
    # Output-node hop (node.rs:189-191)
    eng = dspfx.Engine(N, 256)
    dm = torch_cuda.from_numpy(mix.copy()).cuda()
    eng.mix_finish(dm, 256, N)
    torch_cuda.cuda.synchronize()
    assert np.array_equal(dm.cpu().numpy(), (mix / O.link_divisor(N)).astype(F))


def test_pipelined_mix_bus_matches_inline(dspfx, torch_cuda):
    """process_partials + mix_collect on a second stream == process(mix=...) bit for bit."""
    N, B, blocks = 2048 + 64, 128, 6
    ch = chain5(dspfx, delay=256)
    x = noise_block(N, B * blocks)
    y_ref, mix_ref = run_gpu(dspfx, torch_cuda, ch, x, 3, want_mix=True)
    eng = dspfx.Engine(N, B)
    eng.set_chain(ch)
    dx = torch_cuda.from_numpy(x).cuda()
    dy = torch_cuda.empty_like(dx)
    dm = torch_cuda.zeros(B * blocks, dtype=torch_cuda.float32, device="cuda")
    side = torch_cuda.cuda.Stream()
    a = torch_cuda.cuda.current_stream().cuda_stream
    for b in range(blocks):
        eng.process_partials(dx[b * B:(b + 1) * B], out=dy[b * B:(b + 1) * B], n_frames=B, stream=a)
        eng.mix_collect(dm[b * B:(b + 1) * B], B, stream=side.cuda_stream)
    torch_cuda.cuda.synchronize()
    assert np.array_equal(dy.cpu().numpy(), y_ref)
    assert np.array_equal(dm.cpu().numpy().view(np.uint32), mix_ref.view(np.uint32))
    # protocol errors are reported, not UB
    eng.process_partials(dx[:B], out=dy[:B], n_frames=B)
    with pytest.raises(dspfx.DspfxError):
        eng.process_partials(dx[:B], out=dy[:B], n_frames=B)
    eng.mix_collect(dm[:B], B)
    with pytest.raises(dspfx.DspfxError):
        eng.mix_collect(dm[:B], B)


def test_process_host_path(dspfx, torch_cuda):
    ch = chain3(dspfx, delay=128)
    x = noise_block(96, 128)
    eng = dspfx.Engine(96, 128)
    eng.set_chain(ch)
    y, mix = eng.process_host(x, want_mix=True)
    ref = run_oracle(ch, x, 3)
    assert ulp_diff(y, ref).max() <= 1
    assert np.allclose(mix, ref.astype(np.float64).sum(axis=1), rtol=1e-5, atol=1e-4)


def test_noise_fill_matches_oracle(dspfx, torch_cuda):
    eng = dspfx.Engine(300, 128, channel_offset=12345)
    d = torch_cuda.empty((128, 300), dtype=torch_cuda.float32, device="cuda")
    eng.fill_noise(d, 128, 1000)
    torch_cuda.cuda.synchronize()
    assert np.array_equal(d.cpu().numpy(), noise_block(300, 128, c0=12345, n0=1000))


def test_error_behaviour(dspfx, torch_cuda):
    eng = dspfx.Engine(64, 128)
    with pytest.raises(dspfx.DspfxError):
        eng.set_chain([dspfx.NodeSpec(42)])
    with pytest.raises(dspfx.DspfxError):
        eng.set_chain([dspfx.Reverb(delay_samples=64)])          # reverb.rs:58 clamps to >= 128
    with pytest.raises(dspfx.DspfxError):
        eng.set_chain([dspfx.Distort(1.0, 9)])
    eng.set_chain([dspfx.Gain()])
    t = torch_cuda.zeros((256, 64), device="cuda")
    with pytest.raises(dspfx.DspfxError):
        eng.process(t, n_frames=256)                              # > max_frames
    with pytest.raises(dspfx.DspfxError):
        eng.set_param(3, 0, 1.0)
    with pytest.raises(dspfx.DspfxError):
        dspfx.Engine(0)


@pytest.mark.parametrize("tile", [64, 256])
def test_channel_tiled_layout(dspfx, torch_cuda, tile):
    """The channel-tiled HBM layout [N/W][B][W] gives the same bits as frame-major for every kernel."""
    N = 512
    x, s = noise_block(N, 512), noise_block(N, 512, seed=99)
    cases = [(chain5(dspfx, delay=256), None), (chain3(dspfx, delay=128), None),
             ([dspfx.Gain(0.9), dspfx.Distort(3.0, dspfx.FUZZ), dspfx.BiQuad()], None),
             ([dspfx.Gain(0.5), dspfx.Mix(0.3), dspfx.Reverb(delay_samples=128, decay=0.25)], s),
             ([dspfx.LowPass(0.4), dspfx.Fir(fir_taps(100)), dspfx.HighPass(0.2)], None)]
    for ch, side in cases:
        for blk in (128, 256):
            y0, m0 = run_gpu(dspfx, torch_cuda, ch, x, 3, block=blk, side=side, want_mix=True)
            y1, m1 = run_gpu(dspfx, torch_cuda, ch, x, 3, block=blk, side=side, want_mix=True, tile=tile)
            assert np.array_equal(y0.view(np.uint32), y1.view(np.uint32)), (tile, blk, [n.kind for n in ch])
            assert np.allclose(m0, m1, rtol=1e-5, atol=1e-4)
    # noise fill honours the layout too
    eng = dspfx.Engine(N, 128, channel_offset=7, tile_channels=tile)
    d = torch_cuda.empty(128 * N, dtype=torch_cuda.float32, device="cuda")
    eng.fill_noise(d, 128, 5)
    torch_cuda.cuda.synchronize()
    assert np.array_equal(dspfx.from_layout(d.cpu().numpy(), 128, N, tile), noise_block(N, 128, c0=7, n0=5))
    with pytest.raises(dspfx.DspfxError):
        dspfx.Engine(100, 128, tile_channels=64)          # N % W != 0
    with pytest.raises(dspfx.DspfxError):
        dspfx.Engine(96, 128, tile_channels=96)           # not a power of two


# -------------------------------------------------------------------------- FIR

def test_fir_identity_and_small(dspfx, torch_cuda):
    x = noise_block(64, 256)
    y = run_gpu(dspfx, torch_cuda, [dspfx.Fir()], x, 0)
    assert np.array_equal(y, x)                                              # taps [1.0]
    rng = np.random.default_rng(3)
    xi = rng.integers(-8, 8, (256, 64)).astype(F)
    h = [1, 2, 3, 4]
    ch = [dspfx.Fir(h)]
    y, ref = run_gpu(dspfx, torch_cuda, ch, xi, 0), run_oracle(ch, xi, 0)
    assert np.array_equal(y, ref)                                            # exact incl. the warm-up quirk
    ch = [dspfx.Fir(h, dspfx.FIR_AVERAGE)]
    assert np.array_equal(run_gpu(dspfx, torch_cuda, ch, xi, 0), run_oracle(ch, xi, 0))


def fir_sweep(monkeypatch, which):
    """Name the steady-state FIR sweep a test means: "f32" (fir_skew_kernel / fir_mfma_kernel), "split" (bf16 x 3) or "half"
    (two-part f16 with its bf16 x 3 second pass: the product default)."""
    monkeypatch.setenv("DSPFX_FIR_SPLIT", "0" if which == "f32" else "1")
    if which == "split":
        monkeypatch.setenv("DSPFX_FIR_HALF", "0")
    else:
        monkeypatch.delenv("DSPFX_FIR_HALF", raising=False)


def fir_rel_rms(y, ref):
    err = y.astype(np.float64) - ref.astype(np.float64)
    return np.sqrt(np.mean(err ** 2)) / np.sqrt(np.mean(ref.astype(np.float64) ** 2))


@pytest.mark.parametrize("kernel", ["0", "1", "split", "half"])
@pytest.mark.parametrize("T", [3, 16, 100, 128, 129, 512])
def test_fir_random_vs_oracle(dspfx, torch_cuda, monkeypatch, T, kernel):
    """The FIR kernels (0 = exact f64 VALU, 1 = MFMA with the f32 sweep, split = MFMA with the default split-precision sweep)
    through the warm-up quirk and steady state."""
    monkeypatch.setenv("DSPFX_FIR_KERNEL", "1" if kernel in ("split", "half") else kernel)
    if kernel != "0":
        fir_sweep(monkeypatch, "f32" if kernel == "1" else kernel)
    x = noise_block(64, 128 * 8)
    ch = [dspfx.Gain(0.9), dspfx.Fir(fir_taps(T)), dspfx.Gain(1.1)]
    y, ref = run_gpu(dspfx, torch_cuda, ch, x, 3), run_oracle(ch, x, 3)
    assert fir_rel_rms(y, ref) < FIR_RMS_TOL
    if kernel == "0":
        # f64 accumulation in deque order, split at the VecDeque's wrap point like the reference (fir.rs:201-216):
        # the exact kernel reproduces the oracle bit for bit, gains and hops included
        assert np.array_equal(y.view(np.uint32), ref.view(np.uint32))


@pytest.mark.parametrize("N,block", [(1, 128), (31, 128), (33, 64), (100, 256), (96, 32), (64, 100)])
def test_fir_mfma_ragged_shapes(dspfx, torch_cuda, monkeypatch, N, block):
    """Channel counts off the 32-channel MFMA tile, blocks shorter/longer than 128 frames."""
    monkeypatch.setenv("DSPFX_FIR_KERNEL", "1")
    T = 200
    x = noise_block(N, block * 6)
    ch = [dspfx.Fir(fir_taps(T))]
    y, ref = run_gpu(dspfx, torch_cuda, ch, x, 0, block=block), run_oracle(ch, x, 0)
    assert fir_rel_rms(y, ref) < FIR_RMS_TOL
    assert np.abs(y - ref).max() < 2e-5


@pytest.mark.parametrize("skew", ["1", "0", "split", "half"])
@pytest.mark.parametrize("T", [300, 17, 64, 500, 1000])
def test_fir_mfma_integer_exact(dspfx, torch_cuda, monkeypatch, skew, T):
    """Integer taps and samples are exact in f32: the MFMA path must equal the oracle bit for bit,
    including every warm-up output (fir.rs:193-214 pairs state[k] with taps[k] while filling).
    skew = 1 (default): the steady-state kernel whose output tiles share one set of weights per iteration
    (fir_skew_kernel); 0: the rectangular sweep throughout; split: the split-precision sweep (fir_split_kernel: every
    operand as three bf16 parts, six bf16 MFMAs per 16 taps -- small integers have one part, so it is exact too)."""
    monkeypatch.setenv("DSPFX_FIR_KERNEL", "1")
    if skew in ("split", "half"):                  # (half: samples of magnitude up to 8 are beyond f16's range at the sweep's
        fir_sweep(monkeypatch, skew)               #  scale -- its second pass redoes those tiles on the bf16 x 3 sweep: exact either way)
    else:
        fir_sweep(monkeypatch, "f32")
        monkeypatch.setenv("DSPFX_FIR_SKEW", skew)
    rng = np.random.default_rng(5)
    h = rng.integers(-4, 5, T).astype(np.float64)
    xi = rng.integers(-8, 9, (128 * 5 + (T // 128) * 128, 96)).astype(F)
    for mode in (dspfx.FIR_BALANCED, dspfx.FIR_AVERAGE):
        ch = [dspfx.Fir(h, mode)]
        y, ref = run_gpu(dspfx, torch_cuda, ch, xi, 0), run_oracle(ch, xi, 0)
        assert np.array_equal(y, ref)


@pytest.mark.fir_f32
@pytest.mark.parametrize("block", [128, 256, 100, 64, 48, 16])
@pytest.mark.parametrize("T", [33, 400])
def test_fir_skewed_sweep_block_sizes_integer_exact(dspfx, torch_cuda, monkeypatch, T, block):
    """Blocks that are not 128 frames: slices of 128 + a ragged rest, or short blocks (one wave's two tiles), at both
    tiles-per-wave forms of the skewed kernel; integer data, bit for bit, ten blocks past the warm-up."""
    monkeypatch.setenv("DSPFX_FIR_KERNEL", "1")
    rng = np.random.default_rng(11)
    h = rng.integers(-4, 5, T).astype(np.float64)
    n = ((T + block - 1) // block + 10) * block
    xi = rng.integers(-8, 9, (n, 70)).astype(F)
    ch = [dspfx.Fir(h)]
    y, ref = run_gpu(dspfx, torch_cuda, ch, xi, 0, block=block), run_oracle(ch, xi, 0)
    assert np.array_equal(y, ref)


def test_fir_config4_taps_small_n(dspfx, torch_cuda):
    """BASELINE config 4's filter (4096 taps, seeded decaying noise) at a channel count the oracle
    finishes in seconds; > 32 blocks so the history is full (steady state) at the end."""
    T, N, blocks = 4096, 32, 40
    x = noise_block(N, 128 * blocks)
    ch = [dspfx.Fir(fir_taps(T))]
    y, ref = run_gpu(dspfx, torch_cuda, ch, x, 3), run_oracle(ch, x, 3)
    assert fir_rel_rms(y, ref) < FIR_RMS_TOL
    assert fir_rel_rms(y[-512:], ref[-512:]) < FIR_RMS_TOL      # steady state alone
    assert fir_rel_rms(y[:2048], ref[:2048]) < FIR_RMS_TOL      # warm-up alone


def test_the_default_fir_sweep_is_the_split_one_and_meets_the_bar(dspfx, torch_cuda):
    """A host that asks for nothing gets the two-part f16 sweep in steady state (whole 128-frame blocks, tables fit the
    LDS), the f32 sweep where it does not apply (short slices), the stated tolerance either way, integer data bit for bit
    (its samples of magnitude > 3 go through the sweep's bf16 x 3 second pass); DSPFX_FIR_PRECISION_F32 is the opt-out."""
    T, N, blocks = 1024, 64, 20
    x = noise_block(N, 128 * blocks)
    ch = [dspfx.Fir(fir_taps(T))]
    eng = dspfx.Engine(N, 128, link_flags=3)
    eng.set_chain(ch)
    y = _run_fir_blocks(dspfx, torch_cuda, eng, x)
    assert "fir_halfp_kernel" in eng.describe(), eng.describe()
    ref = run_oracle(ch, x, 3)
    assert fir_rel_rms(y[T:], ref[T:]) < FIR_RMS_TOL
    _run_fir_blocks(dspfx, torch_cuda, eng, x[:64], block=64)            # a 64-frame slice: the f32 sweep serves
    assert "fir_skew_kernel" in eng.describe(), eng.describe()
    eng.set_fir_precision(0, dspfx.FIR_PRECISION_F32)
    _run_fir_blocks(dspfx, torch_cuda, eng, x[:128])
    assert "fir_skew_kernel" in eng.describe(), eng.describe()
    rng = np.random.default_rng(3)
    h = rng.integers(-4, 5, 300).astype(np.float64)
    xi = rng.integers(-8, 9, (128 * 6, 96)).astype(F)
    ei = dspfx.Engine(96, 128, link_flags=0)
    ei.set_chain([dspfx.Fir(h)])
    yi = _run_fir_blocks(dspfx, torch_cuda, ei, xi)
    assert "fir_halfp_kernel" in ei.describe()
    assert np.array_equal(yi, run_oracle([dspfx.Fir(h)], xi, 0))


def test_fir_two_part_f16_sweep_range_and_accuracy_per_channel(dspfx, torch_cuda):
    """fir_half_kernel: f32 operands as f16 hi + f16 lo, three products per term.  f16's range is narrow, so the sweep lists
    the tiles it cannot serve and the bf16 x 3 sweep redoes them right behind it.  Config 4's filter, 96 channels in three
    tiles whose channels are, per CHANNEL: ordinary noise; loud (peaks of 6: beyond 3.998); quiet (2^-15: below the 2^-13
    floor); silent; noise that fades from full scale into the floor and comes back; a huge finite sample (exact kernel).
    The bar holds for EVERY channel on its own -- relative RMS over its own output -- not only in aggregate, block after
    block as channels cross the thresholds in both directions."""
    T, N, blocks = 4096, 96, 44
    x = noise_block(N, 128 * blocks)
    x[:, 3] *= 6.0                                   # tile 0: one loud channel -> the whole tile takes the second pass
    x[:, 40] *= F(2.0 ** -15)                        # tile 1: one quiet channel
    x[:, 41] = 0.0                                   #         and a silent one (fine for the fast path by itself)
    env = np.ones(128 * blocks, F)
    env[128 * 6:128 * 24] = np.logspace(0, -7, 128 * 18).astype(F)
    env[128 * 24:128 * 34] = F(1e-7)
    x[:, 70] *= env                                  # tile 2: fades from full scale to 1e-7 and back: fast -> second pass -> fast
    x[128 * 40 + 9, 71] = 3.0e38                     #         and a sample the bf16 split cannot hold either: the exact kernel
    ch = [dspfx.Fir(fir_taps(T))]
    eng = dspfx.Engine(N, 128, link_flags=0)
    eng.set_chain(ch)
    eng.set_fir_precision(0, dspfx.FIR_PRECISION_HALF)
    y = _run_fir_blocks(dspfx, torch_cuda, eng, x)
    assert "fir_halfp_kernel" in eng.describe(), eng.describe()
    ref = run_oracle(ch, x, 0)
    assert np.array_equal(np.isfinite(y), np.isfinite(ref))
    steady = slice(T, None)
    for c in range(N):
        if c == 71:
            continue
        r = ref[steady, c].astype(np.float64)
        e = y[steady, c].astype(np.float64) - r
        if c == 41:
            assert not y[:, c].any()
            continue
        # (the fading channel: judged over windows in which it has a level of its own, as the bar is a relative one)
        for w in range(0, len(r), 128 * 4):
            rw, ew = r[w:w + 128 * 4], e[w:w + 128 * 4]
            assert np.sqrt(np.mean(ew ** 2)) <= FIR_RMS_TOL * np.sqrt(np.mean(rw ** 2)) + 1e-30, (c, w, np.sqrt(np.mean(ew ** 2)) / np.sqrt(np.mean(rw ** 2)))
    fin = np.isfinite(ref[:, 71])
    assert np.allclose(y[fin, 71], ref[fin, 71], rtol=1e-5, atol=1e-5 * np.abs(ref[fin, 71]).max())
    # the same blocks on the bf16 x 3 sweep alone and on the f32 sweep: the three agree to the bar as well
    for prec in (dspfx.FIR_PRECISION_SPLIT, dspfx.FIR_PRECISION_F32):
        e2 = dspfx.Engine(N, 128, link_flags=0)
        e2.set_chain(ch)
        e2.set_fir_precision(0, prec)
        y2 = _run_fir_blocks(dspfx, torch_cuda, e2, x)
        ok = [c for c in range(N) if c not in (41, 71)]
        assert fir_rel_rms(y2[steady][:, ok], ref[steady][:, ok]) < FIR_RMS_TOL


def test_fir_split_precision_sweep_config4_accuracy(dspfx, torch_cuda, monkeypatch):
    """The opt-in split-precision sweep (DSPFX_FIR_SPLIT=1) on config 4's filter: within the same stated tolerance as the
    f32 sweep (measured 2.9e-7 against 3.3e-7), it really is the kernel that ran, huge finite samples (whose bf16 part
    would round to inf) are routed to the exact kernel like non-finite ones."""
    T, N, blocks = 4096, 32, 40
    x = noise_block(N, 128 * blocks)
    x[128 * 36 + 7, 5] = 3.0e38
    ch = [dspfx.Fir(fir_taps(T))]
    eng = dspfx.Engine(N, 128, link_flags=3)
    eng.set_chain(ch)
    eng.set_fir_precision(0, dspfx.FIR_PRECISION_SPLIT)           # through the C ABI (dspfx_set_fir_precision), not the environment
    with pytest.raises(dspfx.DspfxError):
        eng.set_fir_precision(0, 7)
    y = _run_fir_blocks(dspfx, torch_cuda, eng, x)
    assert "fir_split_kernel" in eng.describe(), eng.describe()
    eng.set_fir_precision(0, dspfx.FIR_PRECISION_F32)
    _run_fir_blocks(dspfx, torch_cuda, eng, x[:128])
    assert "fir_skew_kernel" in eng.describe(), eng.describe()
    ref = run_oracle(ch, x, 3)
    ok = np.ones(N, bool)
    ok[5] = False
    assert fir_rel_rms(y[-512:, ok], ref[-512:, ok]) < FIR_RMS_TOL / 2
    assert np.array_equal(np.isfinite(y), np.isfinite(ref))
    fin = np.isfinite(ref[:, 5])
    assert np.allclose(y[fin, 5], ref[fin, 5], rtol=1e-5, atol=1e-5 * np.abs(ref[fin, 5]).max())


@pytest.mark.parametrize("tile", [0, 256])
def test_fir_fill_phase_is_bit_exact(dspfx, torch_cuda, monkeypatch, tile):
    """While a deque that started empty fills, output n is the prefix sum over samples 0..n of state[m] * taps[m]
    (fir.rs:193-214: nothing is popped, state[k] pairs with taps[k]): fir_warm_scan_kernel keeps that running f64 sum per
    channel -- the same additions as the reference's fresh sum, so the first T outputs equal the oracle's bit for bit on the
    default (MFMA) path, noise data, mix bus included; the blocks after them are the sweeps' (tolerance).  A tap reload
    or DSPFX_FIR_SCAN=0 sends the fill phase to the warm-up sweep instead."""
    T, N, B = 1000, 512 if tile else 70, 128
    x = noise_block(N, B * 12)
    ch = [dspfx.Fir(fir_taps(T), dspfx.FIR_AVERAGE)]
    eng = dspfx.Engine(N, B, link_flags=3, tile_channels=tile)
    eng.set_chain(ch)
    ys, kinds, mixes = [], [], []
    for k in range(12):
        dx = torch_cuda.from_numpy(dspfx.to_layout(x[k * B:(k + 1) * B], tile)).cuda()
        dy, dm = torch_cuda.empty_like(dx), torch_cuda.empty(B, device="cuda")
        eng.process(dx, out=dy, mix=dm, n_frames=B)
        torch_cuda.cuda.synchronize()
        ys.append(dspfx.from_layout(dy.cpu().numpy(), B, N, tile))
        mixes.append(dm.cpu().numpy())
        kinds.append([l for l in eng.describe().splitlines() if l.startswith("stage 0")][0])
    y, ref = np.concatenate(ys), run_oracle(ch, x, 3)
    full = (T // B) * B                                              # whole blocks inside the fill phase
    assert all("fir_warm_scan_kernel" in k for k in kinds[:T // B]) and "fir_warm_scan_kernel" not in kinds[-1], kinds
    assert np.array_equal(y[:full].view(np.uint32), ref[:full].view(np.uint32))
    assert fir_rel_rms(y[full:], ref[full:]) < FIR_RMS_TOL
    want = y.astype(np.float64).sum(axis=1)
    assert np.allclose(np.concatenate(mixes), want, rtol=1e-5, atol=1e-4 * np.abs(want).max())
    monkeypatch.setenv("DSPFX_FIR_SCAN", "0")
    eng2 = dspfx.Engine(N, B, link_flags=3, tile_channels=tile)
    eng2.set_chain(ch)
    y2 = _run_fir_blocks(dspfx, torch_cuda, eng2, x[:B * 2]) if not tile else None
    if y2 is not None:
        assert "fir_mfma_kernel" in eng2.describe() and fir_rel_rms(y2, ref[:B * 2]) < FIR_RMS_TOL


def test_fir_state_export_import(dspfx, torch_cuda):
    T, N = 130, 70
    x = noise_block(N, 128 * 6)
    ch = [dspfx.Fir(fir_taps(T))]
    full = run_gpu(dspfx, torch_cuda, ch, x, 0)
    a = dspfx.Engine(N, 128, link_flags=0)
    a.set_chain(ch)
    dx = torch_cuda.from_numpy(x).cuda()
    dy = torch_cuda.empty_like(dx)
    for f0 in range(0, 384, 128):
        a.process(dx[f0:f0 + 128], out=dy[f0:f0 + 128])
    st = a.state_export(0)
    assert len(st) == 32 + T * N * 4                             # header + the deque as it stands (steady state: T samples)
    assert st[:16].view(np.uint64).tolist() == [384, T]
    hist = st[32:].view(np.float32).reshape(T, N)
    assert np.array_equal(hist, x[384 - T:384])                  # oldest first
    b = dspfx.Engine(N, 128, link_flags=0)
    b.set_chain(ch)
    b.state_import(0, st)
    for f0 in range(384, 768, 128):
        b.process(dx[f0:f0 + 128], out=dy[f0:f0 + 128])
    torch_cuda.cuda.synchronize()
    assert np.allclose(dy.cpu().numpy()[384:], full[384:], rtol=0, atol=2e-6)


@pytest.mark.parametrize("packed", ["1", "0"])
def test_fir_two_part_sweep_quiet_window_right_behind_a_loud_epoch(dspfx, torch_cuda, monkeypatch, packed):
    """the packed sweep takes its window's peak from per-epoch (128 sample times) peaks; with T = 200 the window of the
    block at n0 starts at n0 - 208, 48 samples into an epoch.  Loud samples that end INSIDE those 48 (outside the window) must not
    count for the lower bound 2^-13: the window itself holds only samples of 2^-24, whose f16 lo parts are subnormal (14 significant
    bits instead of 22), so the tile has to be redone by the bf16 x 3 pass -- as fir_half_kernel (DSPFX_FIR_PACKED=0), which
    measures the window it sweeps, always did.  A decaying stream that crosses 2^-13 rides along.  Stated bar: relative RMS
    < 1e-6 in EVERY block against the oracle's f64 accumulation (fir.rs:201-216)."""
    monkeypatch.setenv("DSPFX_FIR_PACKED", packed)
    rng = np.random.default_rng(5)
    N, T, B, m = 64, 200, 128, 6
    h = rng.uniform(-1.0, 1.0, T) * np.exp(-np.arange(T) / 40.0)
    nblocks = m + 8
    x = rng.uniform(-1.0, 1.0, (nblocks * B, N)).astype(F)
    L = 128 * m + 40                                    # the last loud sample is L - 1: 40 samples into epoch m
    x[L:] *= F(2.0 ** -24)
    x[:, 32:] = (x[:, 32:] * (0.5 * np.exp(-np.arange(nblocks * B) / 150.0))[:, None]).astype(F)       # channels 32..63: a fade through 2^-13 and on
    eng = dspfx.Engine(N, B, link_flags=0)
    eng.set_chain([dspfx.Fir(h)])
    eng.set_fir_precision(0, dspfx.FIR_PRECISION_HALF)
    got = _run_fir_blocks(dspfx, torch_cuda, eng, x, B).astype(np.float64)
    assert ("fir_halfp_kernel" if packed == "1" else "fir_half_kernel") in eng.describe(), eng.describe()
    ref = O.run_channels([dspfx.Fir(h).oracle_desc()], x, 0).astype(np.float64)
    worst = 0.0
    for half in (slice(0, 32), slice(32, 64)):
        for k in range(nblocks):
            r, g = ref[k * B:(k + 1) * B, half], got[k * B:(k + 1) * B, half]
            rel = float(np.sqrt(np.mean((g - r) ** 2)) / np.sqrt(np.mean(r ** 2)))
            worst = max(worst, rel)
            assert rel < 1e-6, (packed, "block", k, "channels", half, rel)
    # the block in question: its window [128 (m + 2) - 208, 128 (m + 2) + 127] holds no loud sample, its oldest epoch does
    k = m + 2
    assert np.abs(x[128 * k - 208:128 * (k + 1), :32]).max() < 2.0 ** -23 and np.abs(x[128 * m:128 * k - 208, :32]).max() > 0.5
    eng.close()


@pytest.mark.parametrize("packed", ["1", "0"])
def test_fir_packed_history_stays_in_step_with_the_f32_ring(dspfx, torch_cuda, monkeypatch, packed):
    """Round 5: the two-part f16 sweep reads a copy of the history that the append pass has ALREADY split into f16 hi / lo
    (FirState::ringh + a per-channel peak per 128 sample times) instead of splitting the f32 ring 33 times over.  Only that
    append pass maintains the copy; everything else that writes history -- a spell on another sweep, ragged blocks that start
    and end inside a half chunk, a tap reload that re-bases the ring, state import, reset, placement tuning's park / unpark --
    must leave the two in step (or rebuild the copy).  One stream of blocks through all of them against the oracle, with the
    integer part of the data bit for bit; DSPFX_FIR_PACKED=0 is round 4's sweep on the same stream."""
    monkeypatch.setenv("DSPFX_FIR_PACKED", packed)
    name = "fir_halfp_kernel" if packed == "1" else "fir_half_kernel"
    rng = np.random.default_rng(21)
    N, T = 70, 700
    h = rng.integers(-4, 5, T).astype(np.float64)
    nodes = [O.Node(O.FIR, taps_reversed=h[::-1]) for _ in range(N)]
    eng = dspfx.Engine(N, 128, link_flags=0)
    eng.set_chain([dspfx.Fir(h)])
    eng.set_fir_precision(0, dspfx.FIR_PRECISION_HALF)
    xi = lambda n: rng.integers(-3, 4, (n, N)).astype(F)              # exact in f16's 11 + 11 bits, inside its range

    def both(x, block=128):
        got = _run_fir_blocks(dspfx, torch_cuda, eng, x, block)
        ref = np.stack([np.concatenate([nodes[c].process(x[i:i + block, c]) for i in range(0, len(x), block)]) for c in range(N)], axis=1)
        assert np.array_equal(got.view(np.uint32), ref.view(np.uint32)), np.abs(got - ref).max()

    both(xi(128 * 8))                                                   # fill phase, then steady state
    assert name in eng.describe(), eng.describe()
    both(xi(100 * 5), block=100)                                        # ragged: blocks start / end inside a half chunk and an epoch
    assert name in eng.describe(), eng.describe()
    eng.set_fir_precision(0, dspfx.FIR_PRECISION_F32)                   # a spell on another sweep: the copy goes stale ...
    both(xi(128 * 3))
    assert "fir_skew_kernel" in eng.describe()
    eng.set_fir_precision(0, dspfx.FIR_PRECISION_HALF)                  # ... and is rebuilt from the f32 ring
    both(xi(128 * 2))
    assert name in eng.describe(), eng.describe()
    h2 = rng.integers(-4, 5, 1500).astype(np.float64)                   # a longer impulse response: the ring is re-based
    eng.set_taps(0, h2)
    for nd in nodes:
        nd.set_taps(h2[::-1])
    both(xi(128 * 14))
    assert name in eng.describe(), eng.describe()
    st = eng.state_export(0)                                            # out and in again: the copy is rebuilt from the imported rows
    eng.reset()
    eng.state_import(0, st)
    both(xi(128 * 3))
    x_t = torch_cuda.zeros((128, N), device="cuda")                     # placement tuning parks and restores the FIR history
    eng.tune_placement(x_t, torch_cuda.empty_like(x_t), 128)
    both(xi(128 * 3))
    eng.reset()                                                         # fresh nodes
    nodes = [O.Node(O.FIR, taps_reversed=h2[::-1]) for _ in range(N)]
    both(xi(128 * 14))
    assert name in eng.describe(), eng.describe()
    eng.close()


def _run_fir_blocks(dspfx, torch_cuda, eng, x, block=128):
    y = np.empty_like(x)
    for f0 in range(0, x.shape[0], block):
        dx = torch_cuda.from_numpy(np.ascontiguousarray(x[f0:f0 + block])).cuda()
        dy = torch_cuda.empty_like(dx)
        eng.process(dx, out=dy, n_frames=dx.shape[0])
        torch_cuda.cuda.synchronize()
        y[f0:f0 + block] = dy.cpu().numpy()
    return y


@pytest.mark.parametrize("kernel", ["0", "1"])
def test_fir_exact_kernel_is_bit_exact_and_mfma_within_tolerance(dspfx, torch_cuda, monkeypatch, kernel):
    """The exact kernel accumulates in f64 in deque order AND splits the sum where the reference's VecDeque wraps
    (fir.rs:201-216: two partial sums, each rounded to f32, then added): bit for bit the oracle, warm-up and steady
    state, for tap counts on both sides of the deque's capacity doublings."""
    monkeypatch.setenv("DSPFX_FIR_KERNEL", kernel)
    for T in (1, 2, 5, 16, 31, 64, 100, 256, 300):
        x = noise_block(40, 128 * 6 + 77)
        for mode in (dspfx.FIR_BALANCED, dspfx.FIR_AVERAGE):
            ch = [dspfx.Fir(fir_taps(T), mode)]
            eng = dspfx.Engine(40, 128, link_flags=0)
            eng.set_chain(ch)
            y, ref = _run_fir_blocks(dspfx, torch_cuda, eng, x), run_oracle(ch, x, 0)
            if kernel == "0":
                assert np.array_equal(y.view(np.uint32), ref.view(np.uint32)), (T, mode, ulp_diff(y, ref).max())
            else:
                assert fir_rel_rms(y, ref) < FIR_RMS_TOL, (T, mode)


@pytest.mark.parametrize("kernel", ["0", "1", "rect", "split", "half"])
def test_fir_tap_reload_keeps_the_history(dspfx, torch_cuda, monkeypatch, kernel):
    """dspfx_set_taps = the impulse-response reload of fir.rs:153-171: the taps change, `state` does not.  A history
    longer than the new tap count stays longer (one pop per step, fir.rs:193-197): the output is the new convolution
    delayed by the difference; a shorter one goes on filling front-aligned.  Against the oracle, whose node takes the
    same reloads; integer data is exact on both kernels, random data exact on the f64 kernel."""
    if kernel == "rect":        # the rectangular MFMA sweep in steady state too
        monkeypatch.setenv("DSPFX_FIR_SKEW", "0")
        kernel = "1"
    monkeypatch.setenv("DSPFX_FIR_SPLIT", "0")      # the f32 sweeps unless asked otherwise (the engine's default is the split one)
    if kernel in ("split", "half"):       # the split-precision sweeps in steady state
        fir_sweep(monkeypatch, kernel)
        kernel = "1"
    monkeypatch.setenv("DSPFX_FIR_KERNEL", kernel)
    rng = np.random.default_rng(8)
    N = 70
    for integers in (True, False):
        plan = [(40, 300), (7, 200), (150, 333), (150, 64), (3, 128), (600, 700), (33, 128 * 3)]
        mk = (lambda T: rng.integers(-4, 5, T).astype(np.float64)) if integers else (lambda T: rng.uniform(-1, 1, T))
        x = (rng.integers(-8, 9, (sum(n for _, n in plan), N)).astype(F) if integers
             else noise_block(N, sum(n for _, n in plan)))
        h0 = mk(plan[0][0])
        eng = dspfx.Engine(N, 128, link_flags=0)
        eng.set_chain([dspfx.Fir(h0)])
        nodes = [O.Node(O.FIR, taps_reversed=h0[::-1]) for _ in range(N)]
        f0, got, ref = 0, [], []
        for k, (T, n) in enumerate(plan):
            if k:
                h = mk(T)
                eng.set_taps(0, h)
                for nd in nodes:
                    nd.set_taps(h[::-1])
            seg = x[f0:f0 + n]
            got.append(_run_fir_blocks(dspfx, torch_cuda, eng, seg))
            ref.append(np.stack([np.concatenate([nodes[c].process(seg[i:i + 128, c]) for i in range(0, n, 128)]) for c in range(N)], axis=1))
            f0 += n
        got, ref = np.concatenate(got), np.concatenate(ref)
        if integers or kernel == "0":
            assert np.array_equal(got.view(np.uint32), ref.view(np.uint32)), (integers, np.abs(got - ref).max())
        else:
            assert fir_rel_rms(got, ref) < FIR_RMS_TOL
    # the delay really is there: 40 taps -> 7 taps leaves the output 33 samples late
    assert eng.describe()


@pytest.mark.parametrize("kernel", ["0", "1", "rect", "split", "half"])
def test_fir_non_finite_samples_stay_in_their_channel_and_window(dspfx, torch_cuda, monkeypatch, kernel):
    """inf / NaN samples: the reference's sums turn inf / NaN exactly while the sample is inside the deque (T outputs)
    and only in that channel.  The MFMA sweep multiplies the zero corners of its Toeplitz band with the history, where
    0 x inf would poison neighbouring outputs: non-finite samples are zeroed in the operand, flagged per tile, and the
    flagged tiles are recomputed by the exact kernel -- so the inf / NaN pattern equals the oracle's and every finite
    output stays within the FIR tolerance."""
    if kernel == "rect":        # the rectangular MFMA sweep in steady state too
        monkeypatch.setenv("DSPFX_FIR_SKEW", "0")
        kernel = "1"
    monkeypatch.setenv("DSPFX_FIR_SPLIT", "0")      # the f32 sweeps unless asked otherwise (the engine's default is the split one)
    if kernel in ("split", "half"):       # the split-precision sweeps in steady state
        fir_sweep(monkeypatch, kernel)
        kernel = "1"
    monkeypatch.setenv("DSPFX_FIR_KERNEL", kernel)
    T, N, nf = 200, 100, 128 * 8
    x = noise_block(N, nf)
    x[130, 3] = np.inf
    x[131, 3] = -np.inf          # inf - inf inside one window
    x[300, 40] = np.nan
    x[5, 64] = -np.inf           # during the warm-up
    x[700, 99] = np.inf
    for taps in (fir_taps(T), np.r_[fir_taps(T - 50), np.zeros(50)]):      # zero taps: 0 x inf = NaN in the reference too
        ch = [dspfx.Fir(taps)]
        eng = dspfx.Engine(N, 128, link_flags=3)
        eng.set_chain(ch)
        y, ref = _run_fir_blocks(dspfx, torch_cuda, eng, x), run_oracle(ch, x, 3)
        assert np.array_equal(np.isnan(y), np.isnan(ref))
        assert np.array_equal(np.isposinf(y), np.isposinf(ref)) and np.array_equal(np.isneginf(y), np.isneginf(ref))
        fin = np.isfinite(ref)
        assert (~fin).sum() >= 4 * T and fin[:, [0, 1, 2, 4, 41, 63, 65, 98]].all()       # neighbours in the tile: untouched
        err = y[fin].astype(np.float64) - ref[fin].astype(np.float64)
        assert np.sqrt(np.mean(err ** 2)) / np.sqrt(np.mean(ref[fin].astype(np.float64) ** 2)) < FIR_RMS_TOL
        if kernel == "0":
            assert np.array_equal(y[fin].view(np.uint32), ref[fin].view(np.uint32))


@pytest.mark.parametrize("kernel", ["1", "rect", "0", "split", "half"])
@pytest.mark.parametrize("N,tile,B", [(70, 0, 128), (4096 + 256, 256, 128), (300, 0, 256), (96, 0, 48)])
def test_fir_that_ends_the_chain_feeds_the_mix_bus_itself(dspfx, torch_cuda, monkeypatch, kernel, N, tile, B):
    """When the FIR node ends the chain the sweep's epilogue (and the exact kernel) leave the Output node's mix-bus partials
    per 32-channel tile, instead of an empty chain kernel re-reading the output.  The bus must equal the sum over the
    channels of the block the engine returned -- all three forms (dspfx_process with mix, the pipelined form, flush), ragged
    channel counts, blocks of two slices, short blocks, warm-up and steady state, both sweeps and the f64 kernel, and a
    block with non-finite samples, where the fix-up pass recomputes the flagged tile's partials too."""
    if kernel == "rect":
        monkeypatch.setenv("DSPFX_FIR_SKEW", "0")
        kernel = "1"
    if kernel in ("split", "half"):
        fir_sweep(monkeypatch, kernel)
        kernel = "1"
    monkeypatch.setenv("DSPFX_FIR_KERNEL", kernel)
    T, blocks = 100, 6
    x = noise_block(N, B * blocks)
    x[B * 4 + 5, min(40, N - 1)] = np.inf
    chain = [dspfx.Gain(0.7), dspfx.Fir(fir_taps(T))]
    a, b = dspfx.Engine(N, B, tile_channels=tile), dspfx.Engine(N, B, tile_channels=tile)
    a.set_chain(chain)
    b.set_chain(chain)
    assert "fir" in a.describe().lower()
    mixes = [torch_cuda.full((B,), 7.0, device="cuda") for _ in range(blocks)]
    outs, inline = [], []
    for k in range(blocks):
        dx = torch_cuda.from_numpy(dspfx.to_layout(x[k * B:(k + 1) * B], tile)).cuda()
        y0, y1 = torch_cuda.empty_like(dx), torch_cuda.empty_like(dx)
        m = torch_cuda.empty(B, device="cuda")
        a.process(dx, out=y0, mix=m, n_frames=B)
        b.process_mixpipe(dx, y1, mixes[k - 2] if k >= 2 else None, B)
        torch_cuda.cuda.synchronize()
        assert torch_cuda.equal(y0, y1) or np.array_equal(np.isnan(y0.cpu().numpy()), np.isnan(y1.cpu().numpy()))
        outs.append(dspfx.from_layout(y0.cpu().numpy(), B, N, tile))
        inline.append(m.cpu().numpy())
    b.mixpipe_flush(mixes[blocks - 2], mixes[blocks - 1])
    torch_cuda.cuda.synchronize()
    for k in range(blocks):
        want = outs[k].astype(np.float64).sum(axis=1)
        scale = np.abs(outs[k][np.isfinite(outs[k])]).astype(np.float64).sum() / B + 1.0
        for got in (inline[k], mixes[k].cpu().numpy()):
            fin = np.isfinite(want)
            assert np.array_equal(np.isfinite(got), fin), (k, "non-finite pattern")
            assert not fin.any() or np.abs(got[fin] - want[fin]).max() <= 2e-6 * scale, (k, np.abs(got[fin] - want[fin]).max(), scale)
        assert np.array_equal(inline[k].view(np.uint32)[np.isfinite(inline[k])], mixes[k].cpu().numpy().view(np.uint32)[np.isfinite(inline[k])])
    assert not np.isfinite(np.concatenate(inline)).all()        # the inf really went through the bus


# ---- SignalGen: the reference's control source (signal_gen.rs:55-129) ------------------------------

@pytest.mark.parametrize("mode", [0, 1, 2, 3])
def test_signal_gen_modes(dspfx, torch_cuda, mode):
    """A generator replaces the signal (it has no "in" port); its phase clock wraps at every 128-frame
    block end.  Triangle / Square / Constant are exact; Sine differs from glibc sinf by <= 1 ulp before
    the amplitude multiply (<= 2 after)."""
    N, nf = 100, 128 * 6
    x = noise_block(N, nf)
    for freq, amp in ((100.0, 0.5), (12000.0, 1.0), (0.1, -0.3), (19999.0, 0.7), (441.0, -1.0)):
        chain = [dspfx.SignalGen(amp, freq, mode)]
        for block, tile in ((128, 0), (256, 0), (384, 0), (128, 64)):
            n_ch = 128 if tile else N
            got = run_gpu(dspfx, torch_cuda, chain, x[:, :n_ch] if n_ch <= N else noise_block(n_ch, nf), block=block, tile=tile)
            ref = O.run_channels([n.oracle_desc() for n in chain], np.zeros((nf, 1), F))[:, :1]
            assert got.shape[1] == n_ch and np.array_equal(got, np.repeat(got[:, :1], n_ch, axis=1))   # channels are identical
            d = ulp_diff(got[:, :1], ref)
            assert d.max() <= (2 if mode == 0 else 0), (mode, freq, block, tile, d.max())
    # a block that is not a multiple of 128 ends the generator's block with it (the caller's blocks are
    # the reference's blocks): the oracle is run with the same block length
    chain = [dspfx.SignalGen(0.9, 777.0, mode)]
    got = run_gpu(dspfx, torch_cuda, chain, x[:600, :64], block=100)
    ref = O.run_channels([n.oracle_desc() for n in chain], np.zeros((600, 1), F), block=100)
    assert ulp_diff(got[:, :1], ref).max() <= (2 if mode == 0 else 0)
    # ... feeding effects: generator -> gain -> biquad -> delay (block 256 > D: the engine splits on block boundaries)
    chain = [dspfx.SignalGen(0.8, 1234.5, mode), dspfx.Gain(0.5), dspfx.BiQuad(), dspfx.Reverb(delay_samples=128, decay=0.5),
             dspfx.Distort(2.0, dspfx.SOFT_CLIP)]
    for lf in (3, 0):
        got = run_gpu(dspfx, torch_cuda, chain, x[:, :64], link_flags=lf, block=256)
        ref = O.run_channels([n.oracle_desc() for n in chain], np.zeros((nf, 1), F), lf)
        if mode == 0:
            assert np.abs(got[:, :1] - ref).max() <= 1e-6     # 1-ulp sine differences pass through the filters
        else:
            assert ulp_diff(got[:, :1], ref).max() <= 1, (mode, lf)


def test_signal_gen_control_ports_and_state(dspfx, torch_cuda):
    """amplitude / frequency are `as_input` sliders: per-sample modulation + per-channel latch; the clock is
    per-channel state (export / import / reset)."""
    N, B, blocks = 96, 128, 6
    x = noise_block(N, B * blocks)
    ca = (noise_block(N, B * blocks, seed=31) * F(1.3)).astype(F)
    cf = (noise_block(N, B * blocks, seed=32) * F(0.02) - F(0.9)).astype(F)     # around 1 kHz, varies per channel
    ctl_all = {(0, 0): ca, (0, 1): cf}
    for mode in (0, 1, 2, 3):
        chain = [dspfx.SignalGen(0.5, 300.0, mode), dspfx.Gain(0.9)]
        descs = [n.oracle_desc() for n in chain]
        for lf in (3, 0):
            eng = dspfx.Engine(N, B, link_flags=lf)
            eng.set_chain(chain)
            nodes = []
            dev = {k: torch_cuda.from_numpy(v).cuda() for k, v in ctl_all.items()}
            dx = torch_cuda.from_numpy(x).cuda()
            dy = torch_cuda.empty_like(dx)
            ref = np.empty_like(x)

            def run(b, ks):
                sl = slice(b * B, (b + 1) * B)
                eng.process(dx[sl], out=dy[sl], n_frames=B, ctl={k: dev[k][sl] for k in ks} or None)
                ref[sl] = O.run_channels(descs, x[sl], lf, ctl={k: ctl_all[k][sl] for k in ks} or None, nodes_out=nodes)

            run(0, [(0, 0), (0, 1)])
            run(1, [(0, 1)])               # amplitude latched per channel, frequency still modulated
            run(2, [])                     # both latched: every channel keeps its own frequency
            eng.set_param(0, 1, 2000.0)    # slider store overwrites the latch
            for chn in nodes:
                chn[0].set_param(1, 2000.0)
            run(3, [])
            run(4, [(0, 0)])
            torch_cuda.cuda.synchronize()
            got = dy.cpu().numpy()[:5 * B]
            want = ref[:5 * B]
            if mode == 0:
                # phases are bit-identical, sinf differs by <= 1 ulp => <= 2 ulp after * amplitude, <= 3 after the gain
                assert ulp_diff(got, want).max() <= 3, (mode, lf, ulp_diff(got, want).max())
            else:
                assert np.array_equal(got, want), (mode, lf)
            # clock state: canonical [N] f32
            if mode != 3:
                clk = eng.state_export(0).view(np.float32)
                assert clk.shape == (N,)
                assert np.all((clk >= 0) & (clk < 1)) and len(np.unique(clk)) > N // 2   # per-channel frequencies => per-channel clocks
                # import into a fresh engine: block 5 continues bit-identically
                eng2 = dspfx.Engine(N, B, link_flags=lf)
                eng2.set_chain(chain)
                eng2.set_param(0, 1, 2000.0)
                eng2.state_import(0, eng.state_export(0))
                sl = slice(5 * B, 6 * B)
                y1, y2 = torch_cuda.empty_like(dx[sl]), torch_cuda.empty_like(dx[sl])
                # eng holds latched amplitudes; overwrite both sliders on both engines so only the clock matters
                for e_ in (eng, eng2):
                    e_.set_param(0, 0, 0.25)
                    e_.set_param(0, 1, 2000.0)
                    e_.process(dx[sl], out=y1 if e_ is eng else y2, n_frames=B)
                torch_cuda.cuda.synchronize()
                assert torch_cuda.equal(y1, y2)
                eng.reset()
                eng2 = dspfx.Engine(N, B, link_flags=lf)
                eng2.set_chain(chain)
                for e_ in (eng, eng2):
                    e_.set_param(0, 0, 0.25)
                    e_.set_param(0, 1, 2000.0)
                    e_.process(dx[sl], out=y1 if e_ is eng else y2, n_frames=B)
                torch_cuda.cuda.synchronize()
                assert torch_cuda.equal(y1, y2)


def test_signal_gen_as_lfo_for_another_chain(dspfx, torch_cuda):
    """The reference's use of a generator: an LFO patched into a slider port.  Engine A generates the
    LFO on the GPU, engine B consumes it as the control signal of a gain (tremolo) -- no host round trip."""
    N, B, blocks = 128, 128, 4
    x = noise_block(N, B * blocks)
    lfo_chain = [dspfx.SignalGen(1.0, 750.0, dspfx.SIG_TRIANGLE)]
    fx_chain = [dspfx.Gain(1.0), dspfx.BiQuad()]
    a, b = dspfx.Engine(N, B), dspfx.Engine(N, B)
    a.set_chain(lfo_chain)
    b.set_chain(fx_chain)
    dx = torch_cuda.from_numpy(x).cuda()
    lfo, dy = torch_cuda.empty_like(dx), torch_cuda.empty_like(dx)
    for k in range(blocks):
        sl = slice(k * B, (k + 1) * B)
        a.process(dx[sl], out=lfo[sl], n_frames=B)
        b.process(dx[sl], out=dy[sl], n_frames=B, ctl={(0, 0): lfo[sl]})
    torch_cuda.cuda.synchronize()
    lfo_ref = O.run_channels([n.oracle_desc() for n in lfo_chain], np.zeros((B * blocks, 1), F))
    assert np.array_equal(lfo.cpu().numpy(), np.repeat(lfo_ref, N, axis=1))
    ref = O.run_channels([n.oracle_desc() for n in fx_chain], x, ctl={(0, 0): np.repeat(lfo_ref, N, axis=1)})
    assert ulp_diff(dy.cpu().numpy(), ref).max() <= 1


def _every_node(dspfx):
    exact = [dspfx.Gain(0.7), dspfx.BiQuad(), dspfx.LowPass(0.3), dspfx.HighPass(0.3), dspfx.Reverb(delay_samples=128, decay=0.5),
             dspfx.Add(), dspfx.Mix(0.25)]
    exact += [dspfx.Distort(3.0, m) for m in (dspfx.HARD_CLIP, dspfx.SOFT_CLIP, dspfx.RECIP_SOFT_CLIP, dspfx.SQUARE, dspfx.CHEBYSHEV4)]
    exact += [dspfx.SignalGen(0.6, 1500.0, m) for m in (dspfx.SIG_TRIANGLE, dspfx.SIG_SQUARE, dspfx.SIG_CONSTANT)]
    exact += [dspfx.Envelope(), dspfx.Envelope(0.0, 40.0), dspfx.Envelope(12.0, 300.0)]
    libm = [dspfx.Distort(3.0, m) for m in (dspfx.TANH, dspfx.SIN, dspfx.ATAN)]
    libm += [dspfx.Overdrive(5.0, 0.5, 0.8), dspfx.Chebyshev(4.0, 2.0), dspfx.SignalGen(0.6, 1500.0, dspfx.SIG_SINE)]
    return exact, libm


def test_every_node_in_main_tail_and_control_port_kernels(dspfx, torch_cuda):
    """Each node kind / mode on its own through every interpreter instantiation: the whole-wave launch, the
    guarded one-wave tail (N % 64 != 0) and -- with a connected control port on a second node -- their
    control-port versions.  (The per-kind tests above mostly use whole waves; this one found a dropped
    branch in the guarded kernel's code.)"""
    N, B, blocks = 100, 128, 3          # 64 channels in the main launch + 36 in the tail
    x, side = noise_block(N, B * blocks), noise_block(N, B * blocks, seed=9)
    ctl = (noise_block(N, B * blocks, seed=77) * F(0.5)).astype(F)
    exact, libm = _every_node(dspfx)
    for node in exact + libm:
        bar = 1 if any(node is e for e in exact) else LIBM_COMPOSITE_ULP
        for with_ctl in (False, True):
            chain = [node, dspfx.Gain(1.0)] if with_ctl else [node]
            cports = {(1, 0): ctl} if with_ctl else None
            eng = dspfx.Engine(N, B, link_flags=3)
            eng.set_chain(chain)
            dx, ds = torch_cuda.from_numpy(x).cuda(), torch_cuda.from_numpy(side).cuda()
            dc = torch_cuda.from_numpy(ctl).cuda()
            dy = torch_cuda.empty_like(dx)
            for b in range(blocks):
                sl = slice(b * B, (b + 1) * B)
                eng.process(dx[sl], out=dy[sl], side=ds[sl], n_frames=B, ctl={(1, 0): dc[sl]} if with_ctl else None)
            torch_cuda.cuda.synchronize()
            ref = O.run_channels([n.oracle_desc() for n in chain], x, 3, side, ctl=cports)
            d = ulp_diff(dy.cpu().numpy(), ref)
            # main launch (channels < 64) and tail (>= 64) judged separately so a failure names the kernel
            assert d[:, :64].max() <= bar, ("main", node.kind, node.mode, with_ctl, d[:, :64].max())
            assert d[:, 64:].max() <= bar, ("tail", node.kind, node.mode, with_ctl, d[:, 64:].max())


def test_envelope_follower(dspfx, torch_cuda):
    """envelope.rs:34-52 (dasp peak detector): gains come from the host's powf like the reference's, the
    recurrence itself is plain f32 => bit-exact against the oracle, alone and as a stage of a chain, with state
    carried across blocks, exported / imported and cleared by reset."""
    N, B, blocks = 200, 128, 5
    x = noise_block(N, B * blocks)
    for att, rel in ((0.0, 0.0), (0.0, 64.0), (5.0, 0.0), (48.0, 1000.0), (0.5, 0.25)):
        chain = [dspfx.Envelope(att, rel)]
        for lf in (3, 0):
            got = run_gpu(dspfx, torch_cuda, chain, x, link_flags=lf)
            assert np.array_equal(got.view(np.uint32), run_oracle(chain, x, lf).view(np.uint32)), (att, rel, lf)
    chain = [dspfx.BiQuad(), dspfx.Envelope(20.0, 400.0), dspfx.Gain(2.0), dspfx.Reverb(delay_samples=128, decay=0.4)]
    for block, tile in ((128, 0), (256, 0), (128, 64)):
        n_ch = 192 if tile else N
        got = run_gpu(dspfx, torch_cuda, chain, np.ascontiguousarray(x[:, :n_ch]), block=block, tile=tile)
        assert ulp_diff(got, run_oracle(chain, np.ascontiguousarray(x[:, :n_ch]))).max() <= 1
    # parameter change between blocks, state export / import, reset
    eng = dspfx.Engine(N, B)
    eng.set_chain([dspfx.Envelope(10.0, 200.0)])
    nodes = []
    dx = torch_cuda.from_numpy(x).cuda()
    dy = torch_cuda.empty_like(dx)
    ref = np.empty_like(x)
    descs = [dspfx.Envelope(10.0, 200.0).oracle_desc()]
    for b in range(blocks):
        sl = slice(b * B, (b + 1) * B)
        if b == 2:
            eng.set_param(0, 1, 20.0)
            for chn in nodes:
                chn[0].set_param(1, 20.0)
        eng.process(dx[sl], out=dy[sl], n_frames=B)
        ref[sl] = O.run_channels(descs, x[sl], 3, nodes_out=nodes)
    torch_cuda.cuda.synchronize()
    assert np.array_equal(dy.cpu().numpy(), ref)
    env = eng.state_export(0).view(np.float32)
    assert env.shape == (N,) and np.array_equal(env, ref[-1])            # the state IS the last output
    eng2 = dspfx.Engine(N, B)
    eng2.set_chain([dspfx.Envelope(10.0, 20.0)])
    eng2.state_import(0, eng.state_export(0))
    y1, y2 = torch_cuda.empty_like(dx[:B]), torch_cuda.empty_like(dx[:B])
    eng.process(dx[:B], out=y1, n_frames=B)
    eng2.process(dx[:B], out=y2, n_frames=B)
    torch_cuda.cuda.synchronize()
    assert torch_cuda.equal(y1, y2)
    eng.reset()
    eng.process(dx[:B], out=y1, n_frames=B)
    torch_cuda.cuda.synchronize()
    assert np.array_equal(y1.cpu().numpy(), O.run_channels([dspfx.Envelope(10.0, 20.0).oracle_desc()], x[:B], 3))


def _random_exact_node(dspfx, rng):
    k = rng.integers(0, 9)      # Fuzz is left out: its block-global mean is summed in another order (test_fuzz's bar)
    if k == 0:
        return dspfx.Gain(float(rng.uniform(0.1, 2.0)))
    if k == 1:
        r, th, a0 = rng.uniform(0.1, 0.95), rng.uniform(0.1, 3.0), rng.uniform(0.5, 2.0)
        return dspfx.BiQuad(a0, -2 * r * np.cos(th) * a0, r * r * a0, *(rng.uniform(-1, 1, 3) * a0))
    if k == 2:
        return dspfx.LowPass(float(rng.uniform(0.0, 1.0)))
    if k == 3:
        return dspfx.HighPass(float(rng.uniform(0.0, 1.0)))
    if k == 4:
        return dspfx.Reverb(delay_samples=int(rng.integers(128, 700)), decay=float(rng.uniform(0.0, 0.9)))
    if k == 5:
        return dspfx.Distort(float(rng.uniform(0.0, 6.0)),
                             int(rng.choice([dspfx.HARD_CLIP, dspfx.SOFT_CLIP, dspfx.RECIP_SOFT_CLIP, dspfx.SQUARE, dspfx.CHEBYSHEV4])))
    if k == 6:
        return dspfx.Add()
    if k == 7:
        return dspfx.Mix(float(rng.uniform(0.0, 1.0)))
    if k == 8:
        return dspfx.Envelope(float(rng.choice([0.0, 3.0, 50.0])), float(rng.choice([0.0, 7.0, 400.0])))
    return dspfx.Gain(1.0)


@pytest.mark.parametrize("seed", range(8))
def test_random_chains_of_exact_nodes(dspfx, torch_cuda, seed):
    """Seeded random chains (1..12 nodes of the exact-arithmetic kinds, random parameters, ragged channel
    counts, both layouts, every link-flag setting, blocks of 128 or 256 frames): the interpreter, its guarded
    tail, stage splitting (> 8 fused nodes) and the delay sub-block split against the oracle, <= 1 ulp."""
    rng = np.random.default_rng(1000 + seed)
    for case in range(3):
        n_nodes = int(rng.integers(1, 13))
        chain = [_random_exact_node(dspfx, rng) for _ in range(n_nodes)]
        tile = int(rng.choice([0, 64]))
        N = int(rng.choice([64, 128, 320])) if tile else int(rng.choice([1, 63, 100, 129, 273]))
        block = int(rng.choice([128, 256]))
        lf = int(rng.choice([0, 1, 3]))
        nf = 768
        x, side = noise_block(N, nf, seed=seed * 7 + case), noise_block(N, nf, seed=seed * 7 + case + 100)
        got = run_gpu(dspfx, torch_cuda, chain, x, link_flags=lf, block=block, side=side, tile=tile)
        ref = run_oracle(chain, x, lf, side)
        ok = np.isfinite(ref)
        assert np.array_equal(np.isfinite(got), ok), (seed, case)
        d = ulp_diff(got[ok], ref[ok])
        assert d.size == 0 or d.max() <= 1, (seed, case, [(n.kind, n.mode) for n in chain], N, block, tile, lf, int(d.max()))
        assert np.array_equal(np.signbit(got[ok]), np.signbit(ref[ok])), (seed, case)     # the sign of zeros too


@pytest.mark.parametrize("N,tile", [(4096 + 64 * 300 + 37, 0), (65536, 256), (100, 0)])
def test_mix_bus_pipelined_inside_the_kernel(dspfx, torch_cuda, N, tile):
    """dspfx_process_mixpipe: block k's launch finishes the mix bus of blocks k-1 / k-2 in its first workgroups.
    Same reduction tree as the stand-alone kernels => bit-identical to dspfx_process(mix) + dspfx_mix_finish,
    delivered two calls late; flush drains the last two blocks; the outputs are untouched."""
    B, blocks = 128, 7
    chain = chain5(dspfx, 256)
    x = noise_block(N, B * blocks)
    ref_eng, eng = dspfx.Engine(N, B, tile_channels=tile), dspfx.Engine(N, B, tile_channels=tile)
    ref_eng.set_chain(chain)
    eng.set_chain(chain)
    for n_conn in (0, N):
        ref_eng.reset()
        eng.reset()
        want_mix, want_out, got_out = [], [], []
        mixes = [torch_cuda.full((B,), 7.0, device="cuda") for _ in range(blocks)]
        for k in range(blocks):
            dx = torch_cuda.from_numpy(dspfx.to_layout(x[k * B:(k + 1) * B], tile)).cuda()
            y0, y1 = torch_cuda.empty_like(dx), torch_cuda.empty_like(dx)
            m = torch_cuda.empty(B, device="cuda")
            ref_eng.process(dx, out=y0, mix=m, n_frames=B)
            if n_conn:
                ref_eng.mix_finish(m, B, n_conn)
            eng.process_mixpipe(dx, y1, mixes[k - 2] if k >= 2 else None, B, n_connected=n_conn)
            torch_cuda.cuda.synchronize()
            want_mix.append(m.cpu().numpy())
            want_out.append(y0.cpu().numpy())
            got_out.append(y1.cpu().numpy())
        eng.mixpipe_flush(mixes[blocks - 2], mixes[blocks - 1], n_connected=n_conn)
        torch_cuda.cuda.synchronize()
        for k in range(blocks):
            assert np.array_equal(got_out[k], want_out[k]), (k, "out")
            assert np.array_equal(mixes[k].cpu().numpy().view(np.uint32), want_mix[k].view(np.uint32)), (N, tile, n_conn, k)
    # a single block in flight, and changing n_frames without a flush is refused
    eng.reset()
    dx = torch_cuda.from_numpy(dspfx.to_layout(x[:B], tile)).cuda()
    y = torch_cuda.empty_like(dx)
    m1 = torch_cuda.empty(B, device="cuda")
    eng.process_mixpipe(dx, y, None, B)
    eng.mixpipe_flush(None, m1)
    ref_eng.reset()
    m0 = torch_cuda.empty(B, device="cuda")
    ref_eng.process(dx, out=y, mix=m0, n_frames=B)
    torch_cuda.cuda.synchronize()
    assert torch_cuda.equal(m0, m1)
    eng.process_mixpipe(dx, y, None, B)
    with pytest.raises(dspfx.DspfxError):
        eng.process_mixpipe(dx, y, None, 64)


@pytest.mark.parametrize("which", ["chain3", "chain5"])
def test_time_sliced_kernel_equals_the_standard_one(dspfx, torch_cuda, monkeypatch, which):
    """Engines of at most 131072 channels run whole 128-frame blocks of the BASELINE chains through chain_ts_kernel: a
    workgroup owns 64 channels, its four waves take 32 frames each, state passes from wave to wave through LDS.  Every
    recurrence still sees its frames in order: outputs, mix bus and carried state must equal the standard kernel bit
    for bit (and the oracle within the chain's bar), in both layouts, with a ragged channel count, across blocks."""
    chain = chain3(dspfx, 300) if which == "chain3" else chain5(dspfx, 300)
    # (65536 channels: the 3-node chain's time-sliced kernel runs two channels per lane there)
    for N, tile in ((64 * 37 + 5, 0), (4096, 256), (8192, 0), (65536, 256)):
        x = noise_block(N, 128 * 5)
        outs = {}
        for ts in ("1", "0"):
            monkeypatch.setenv("DSPFX_VARIANT", "ts=" + ts)
            eng = dspfx.Engine(N, 128, link_flags=3, tile_channels=tile)
            eng.set_chain(chain)
            assert ("time-sliced" in eng.describe()) == (ts == "1"), eng.describe()
            ys, ms = [], []
            for k in range(5):
                nf = 64 if k == 3 else 128        # a short block in between: the standard kernel takes it, state carries on
                dx = torch_cuda.from_numpy(dspfx.to_layout(x[k * 128:k * 128 + nf], tile)).cuda()
                dy, dm = torch_cuda.empty_like(dx), torch_cuda.empty(nf, device="cuda")
                eng.process(dx, out=dy, mix=dm, n_frames=nf)
                torch_cuda.cuda.synchronize()
                ys.append(dspfx.from_layout(dy.cpu().numpy(), nf, N, tile))
                ms.append(dm.cpu().numpy().copy())
            outs[ts] = (ys, ms, eng.state_export(0 if which == "chain5" else 1))
        for a, b in zip(outs["1"][0], outs["0"][0]):
            assert np.array_equal(a.view(np.uint32), b.view(np.uint32))
        for a, b in zip(outs["1"][1], outs["0"][1]):
            # the bus: the same sum, associated differently (the standard kernel adds its four waves' totals into one row
            # per workgroup, the time-sliced kernel's workgroup IS one wave's worth of channels; channels per lane differ
            # above 57344 channels): the bus' own bar, not bit equality
            assert np.allclose(a, b, rtol=2e-5, atol=2e-3)
        assert np.array_equal(outs["1"][2], outs["0"][2])
        monkeypatch.delenv("DSPFX_VARIANT")
        ref = run_oracle(chain, x[:256], 3)
        got = np.concatenate(outs["1"][0][:2])
        assert ulp_diff(got, ref).max() <= 1


def test_mix_allreduce_through_the_c_abi(dspfx, torch_cuda):
    """dspfx_comm_create / dspfx_mix_allreduce: the mix bus' one collective behind the C ABI.  A 1-rank communicator made
    from a real unique id runs RCCL's ncclCommInitRank and ncclAllReduce on this GPU (the sum over one rank is the
    identity), then the Output hop with the global count; parallel.PipelinedMixBus driven through it equals the
    single-GPU bus bit for bit."""
    from dsp_stuff_amd import parallel as P
    N, B = 4096 * 3, 128
    uid = dspfx.comm_unique_id()
    assert len(uid) == dspfx.COMM_ID_BYTES and any(uid)
    comm = dspfx.Comm(0, 1, 0, uid)
    chain = chain5(dspfx, 256)
    ref_eng, eng = dspfx.Engine(N, B), dspfx.Engine(N, B)
    ref_eng.set_chain(chain)
    eng.set_chain(chain)
    blocks = 11
    x = noise_block(N, B * blocks, seed=3)
    want = []
    for k in range(blocks):
        dx = torch_cuda.from_numpy(x[k * B:(k + 1) * B]).cuda()
        y, m, m2 = torch_cuda.empty_like(dx), torch_cuda.empty(B, device="cuda"), torch_cuda.empty(B, device="cuda")
        ref_eng.process(dx, out=y, mix=m, n_frames=B)
        m2.copy_(m)
        ref_eng.mix_finish(m, B, 5 * N)
        eng_tmp_sum = m2                                     # the un-normalised bus of this block
        ref_eng.mix_allreduce(comm, eng_tmp_sum, B, 5 * N)   # direct call: all-reduce over one rank + Output hop
        torch_cuda.cuda.synchronize()
        assert np.array_equal(m.cpu().numpy().view(np.uint32), eng_tmp_sum.cpu().numpy().view(np.uint32))
        want.append(m.cpu().numpy())
    cs, ms = torch_cuda.cuda.Stream(), torch_cuda.cuda.Stream()
    with torch_cuda.cuda.stream(cs):
        pb = P.PipelinedMixBus(eng, 5 * N, B, cs, ms, world=1, batch=4, device="cuda", comm=comm)
        dxs = [torch_cuda.from_numpy(x[k * B:(k + 1) * B]).cuda() for k in range(blocks)]
        y = torch_cuda.empty_like(dxs[0])
        torch_cuda.cuda.synchronize()
        got = {}
        for k in range(blocks):
            pb.step(dxs[k], y)
            if k >= 2 and (k - 2 + 1) % 4 == 0:
                torch_cuda.cuda.synchronize()
                q = (k - 2) // 4
                for j in range(q * 4, (q + 1) * 4):
                    got[j] = pb._row(j).cpu().numpy().copy()
        rows = {j: pb._row(j) for j in range(blocks) if j not in got}
        pb.drain()
        torch_cuda.cuda.synchronize()
        for j, r in rows.items():
            got[j] = r.cpu().numpy().copy()
    for j in range(blocks):
        assert np.array_equal(got[j].view(np.uint32), want[j].view(np.uint32)), j
    comm.close()
    with pytest.raises(dspfx.DspfxError):
        dspfx.Comm(0, 2, 5, uid)                             # rank out of range


@pytest.mark.parametrize("same_block", [False, True])
def test_pipelined_mix_bus_with_batched_collective_path(dspfx, torch_cuda, same_block):
    """parallel.PipelinedMixBus (the multi-GPU form of the in-kernel pipeline; world = 1 here, so the RCCL call
    is skipped but rings, events, batching and drain are the real thing): every block's bus equals
    dspfx_process(mix) + dspfx_mix_finish bit for bit, including the partly filled last batch.  same_block: the rows come
    from dspfx_process_bus (the bus of the launch's own block) instead of the in-kernel pipeline (two calls late)."""
    lag = 0 if same_block else 2
    from dsp_stuff_amd import parallel as P
    N, B, batch = 4096 * 5, 128, 4
    chain = chain5(dspfx, 256)
    ref_eng, eng = dspfx.Engine(N, B), dspfx.Engine(N, B)
    ref_eng.set_chain(chain)
    eng.set_chain(chain)
    cs, ms = torch_cuda.cuda.Stream(), torch_cuda.cuda.Stream()
    for blocks in (1, 2, 3, 9, 13):                 # 9 and 13 leave 1 row / a ring boundary inside the flush
        ref_eng.reset()
        eng.reset()
        x = noise_block(N, B * blocks, seed=blocks)
        want = []
        for k in range(blocks):
            dx = torch_cuda.from_numpy(x[k * B:(k + 1) * B]).cuda()
            y = torch_cuda.empty_like(dx)
            m = torch_cuda.empty(B, device="cuda")
            ref_eng.process(dx, out=y, mix=m, n_frames=B)
            ref_eng.mix_finish(m, B, 3 * N)
            torch_cuda.cuda.synchronize()
            want.append(m.cpu().numpy())
        with torch_cuda.cuda.stream(cs):
            pb = P.PipelinedMixBus(eng, 3 * N, B, cs, ms, world=1, batch=batch, device="cuda", same_block=same_block)
            dxs = [torch_cuda.from_numpy(x[k * B:(k + 1) * B]).cuda() for k in range(blocks)]
            y = torch_cuda.empty_like(dxs[0])
            torch_cuda.cuda.synchronize()
            got = {}
            for k in range(blocks):
                pb.step(dxs[k], y)
                # a ring is only reused three batches later: read completed batches before that happens
                if k >= lag and (k - lag + 1) % batch == 0:
                    torch_cuda.cuda.synchronize()
                    q = (k - lag) // batch
                    for j in range(q * batch, (q + 1) * batch):
                        got[j] = pb._row(j).cpu().numpy().copy()
            rows = {j: pb._row(j) for j in range(blocks) if j not in got}
            pb.drain()
            torch_cuda.cuda.synchronize()
            for j, r in rows.items():
                got[j] = r.cpu().numpy().copy()
            for j, r in pb.results().items():           # what results() still serves equals what was read on the way
                assert np.array_equal(r.cpu().numpy().view(np.uint32), got[j].view(np.uint32)), (blocks, j)
        for k in range(blocks):
            assert np.array_equal(got[k].view(np.uint32), want[k].view(np.uint32)), (blocks, k)


def test_tune_placement_changes_speed_not_results(dspfx, torch_cuda):
    """dspfx_tune_placement re-times the ring groups with the real chain on the caller's buffers and keeps the
    fastest; it PRESERVES the DSP state (filter rows snapshotted, ring groups parked and put back at the same ring
    position) and must not change a single output bit -- called before the first block and in the middle of a run."""
    N, B, blocks = 131072, 128, 5            # 64 MiB ring groups: large enough for the tuner to engage
    chain = chain5(dspfx, 300)               # 3 groups, the last one partly used
    x = noise_block(N, B * blocks)
    eng, ref = dspfx.Engine(N, B, tile_channels=256), dspfx.Engine(N, B, tile_channels=256)
    eng.set_chain(chain)
    ref.set_chain(chain)
    dxs = [torch_cuda.from_numpy(dspfx.to_layout(x[k * B:(k + 1) * B], 256)).cuda() for k in range(blocks)]
    y, y_ref = torch_cuda.empty_like(dxs[0]), torch_cuda.empty_like(dxs[0])
    eng.tune_placement(dxs[0], y, B)
    assert "re-placed" in eng.describe()
    scratch = torch_cuda.empty_like(dxs[0])
    for k in range(blocks):
        eng.process(dxs[k], out=y, n_frames=B)
        ref.process(dxs[k], out=y_ref, n_frames=B)
        torch_cuda.cuda.synchronize()
        assert torch_cuda.equal(y, y_ref), k
        if k in (1, 2, 3):            # mid-run, ring part full (D = 300: the position wraps inside the third group)
            eng.tune_placement(dxs[(k + 2) % blocks], scratch, B)
    with pytest.raises(dspfx.DspfxError):
        eng.tune_placement(dxs[0], None, B)
    eng.process_mixpipe(dxs[0], y, None, B)
    with pytest.raises(dspfx.DspfxError):                       # blocks in the mix pipeline: flush first
        eng.tune_placement(dxs[0], y, B)
    # no large ring: a no-op
    small = dspfx.Engine(4096, B)
    small.set_chain(chain3(dspfx, 256))
    t = torch_cuda.zeros((B, 4096), device="cuda")
    small.tune_placement(t, t, B)


@pytest.mark.parametrize("variant", ["static=0,f=8,cpl=2", "static=0,f=4,cpl=2"])
def test_two_channel_interpreter_on_every_node_and_random_chains(dspfx, torch_cuda, monkeypatch, variant):
    """The interpreter's two-channels-per-lane instantiations (the default above 131072 channels) forced at test
    sizes: every arithmetic node kind on its own, the libm kinds (dyn_libm_f8_c2), and seeded random chains, with a
    ragged channel count so the one-channel guarded tail runs next to it and the mix bus spans both."""
    monkeypatch.setenv("DSPFX_VARIANT", variant)
    N, B, blocks = 128 * 3 + 2 * 17, 128, 3        # 384 channels in the main launch, 34 in the tail
    x, side = noise_block(N, B * blocks), noise_block(N, B * blocks, seed=9)
    exact, libm = _every_node(dspfx)
    for node in exact + libm:
        bar = 1 if any(node is e for e in exact) else LIBM_COMPOSITE_ULP
        got, mix = run_gpu(dspfx, torch_cuda, [node], x, side=side, want_mix=True)
        ref = run_oracle([node], x, 3, side)
        assert ulp_diff(got, ref).max() <= bar, (variant, node.kind, node.mode)
        want_mix = ref.astype(np.float64).sum(axis=1)
        assert np.allclose(mix, want_mix, rtol=1e-4, atol=1e-3 * max(1.0, np.abs(want_mix).max())), (variant, node.kind)
    eng = dspfx.Engine(N, B)
    eng.set_chain([dspfx.Gain(0.5), dspfx.BiQuad()])
    assert "_c2" in eng.describe(), eng.describe()
    rng = np.random.default_rng(77)
    for case in range(6):
        chain = [_random_exact_node(dspfx, rng) for _ in range(int(rng.integers(2, 11)))]
        tile = int(rng.choice([0, 64]))
        n_ch = 448 if tile else N
        xs, ss = np.ascontiguousarray(noise_block(n_ch, B * 4, seed=case)), np.ascontiguousarray(noise_block(n_ch, B * 4, seed=case + 50))
        lf = int(rng.choice([0, 1, 3]))
        got = run_gpu(dspfx, torch_cuda, chain, xs, link_flags=lf, side=ss, tile=tile, block=int(rng.choice([128, 256])))
        ref = run_oracle(chain, xs, lf, ss)
        ok = np.isfinite(ref)
        assert np.array_equal(np.isfinite(got), ok)
        assert ulp_diff(got[ok], ref[ok]).max() <= 1, (variant, case, [(n.kind, n.mode) for n in chain], tile, lf)


def test_fast_f64_tanh_sin_atan_match_the_library_path_exhaustively(dspfx, torch_cuda):
    """The engine's own f64 tanh / sin / atan against the math library's, all 2^32 inputs, on the device."""
    for func, name, allowed in ((0, "tanh", 16), (1, "sin", 0), (2, "atan", 0), (3, "exp", 0)):
        n, worst = dspfx.verify_libm(func)
        assert n <= allowed and worst <= 1, (name, n, worst)
    # Fuzz's per-lane divisions as f64 products (IEEE fallback for subnormal quotients): identical to IEEE division
    # on 2^32 (numerator, hashed divisor) pairs per seed
    for seed in (4, 5):
        assert dspfx.verify_libm(seed) == (0, 0), seed


@pytest.mark.parametrize("jit", ["0", "default"])
def test_control_ports_two_channel_interpreter_above_131072_channels(dspfx, torch_cuda, monkeypatch, jit):
    """Above 131072 channels a chain with connected control ports runs on the two-channels-per-lane control-port
    interpreter (dyn_mod_f8_c2).  Channels are independent, so its output must equal, bit for bit, what smaller
    engines (one channel per lane, validated against the oracle by test_control_ports) produce for channel slices;
    a few channels are also checked against the oracle directly."""
    # jit "0": the control-port interpreter (dyn_mod_f8_c2); "default": the run-time specialised control-port kernel
    if jit == "0":
        monkeypatch.setenv("DSPFX_JIT", "0")
    else:
        monkeypatch.delenv("DSPFX_JIT", raising=False)
    N, B, blocks = 131072 + 1024 + 2, 128, 2          # even, N % 128 == 2: a 2-channel guarded tail rides along
    chain = [dspfx.Gain(1.0), dspfx.Distort(3.0, dspfx.HARD_CLIP), dspfx.Mix(0.5), dspfx.BiQuad(), dspfx.Overdrive(5.0, 0.5, 0.8)]
    keys = [(0, 0), (1, 0), (2, 0), (4, 1)]

    def run(c0, n):
        eng = dspfx.Engine(n, B, channel_offset=c0)
        eng.set_chain(chain)
        outs = []
        for b in range(blocks):
            x = torch_cuda.empty((B, n), device="cuda")
            side = torch_cuda.empty_like(x)
            eng.fill_noise(x, B, b * B, 0x5EED0001)
            eng.fill_noise(side, B, b * B, 5)
            ctl = {}
            for j, k in enumerate(keys):
                t = torch_cuda.empty_like(x)
                eng.fill_noise(t, B, b * B, 100 + j)
                ctl[k] = t
            y = torch_cuda.empty_like(x)
            eng.process(x, out=y, side=side, n_frames=B, ctl=ctl)
            torch_cuda.cuda.synchronize()
            outs.append(y.cpu().numpy())
        return np.concatenate(outs)

    big = run(0, N)
    half = 66048
    small = np.concatenate([run(0, half), run(half, N - half)], axis=1)
    assert np.array_equal(big.view(np.uint32), small.view(np.uint32))
    chans = np.array([0, 1, 127, 128, 70000, N - 3, N - 2, N - 1])
    nf = B * blocks
    x = O.noise(0x5EED0001, chans, np.arange(nf))
    side = O.noise(5, chans, np.arange(nf))
    ctl = {k: O.noise(100 + j, chans, np.arange(nf)) for j, k in enumerate(keys)}
    ref = O.run_channels([n.oracle_desc() for n in chain], x, 3, side, ctl=ctl)
    assert np.abs(big[:, chans] - ref).max() <= 2e-6          # overdrive's atan feeds further f32 ops (test_control_ports' bar)


def test_pipelined_host_path_matches_device_path(dspfx, torch_cuda):
    """From 262144 channels on, dspfx_process_host cuts a block into channel parts and overlaps upload, kernel and
    download of different parts; results must equal the device path bit for bit (ragged tail, side input, two
    fused stages, state and ring carried across blocks)."""
    N, B, blocks = 262144 + 100, 128, 3
    chain = [dspfx.Gain(0.9), dspfx.BiQuad(), dspfx.Mix(0.3), dspfx.Reverb(delay_samples=256, decay=0.5), dspfx.LowPass(0.2),
             dspfx.Distort(2.0, dspfx.SOFT_CLIP), dspfx.HighPass(0.1), dspfx.Envelope(3.0, 50.0), dspfx.Gain(1.1), dspfx.Add()]
    host, dev = dspfx.Engine(N, B), dspfx.Engine(N, B)
    host.set_chain(chain)
    dev.set_chain(chain)
    assert len([l for l in host.describe().splitlines() if l.startswith("stage")]) == 2
    rng = np.random.default_rng(8)
    px, ps, py = dspfx.PinnedArray((B, N)), dspfx.PinnedArray((B, N)), dspfx.PinnedArray((B, N))   # page-locked: the pipelined form
    for b in range(blocks):
        x = rng.uniform(-1, 1, (B, N)).astype(F)
        side = rng.uniform(-1, 1, (B, N)).astype(F)
        px.array[:], ps.array[:] = x, side
        got, got_mix = host.process_host(px.array, side=ps.array, out=py.array, want_mix=True)
        dx, ds = torch_cuda.from_numpy(x).cuda(), torch_cuda.from_numpy(side).cuda()
        dy = torch_cuda.empty_like(dx)
        dm = torch_cuda.empty(B, device="cuda")
        dev.process(dx, out=dy, side=ds, mix=dm, n_frames=B)
        torch_cuda.cuda.synchronize()
        assert np.array_equal(got.view(np.uint32), dy.cpu().numpy().view(np.uint32)), b
        assert np.array_equal(got_mix.view(np.uint32), dm.cpu().numpy().view(np.uint32)), b   # same partials, same reduction tree


def test_default_variant_selection_above_131072_channels_matches_small_engines(dspfx, torch_cuda, monkeypatch):
    """Above 131072 channels the engine picks a run-time specialised kernel or the two-channels-per-lane interpreter
    by itself.  Channels are
    independent, so a big engine must reproduce, bit for bit, what two smaller engines (one channel per lane) give for
    the two halves -- arithmetic chain, libm chain, tiled layout, ragged tail, mix bus."""
    B, blocks = 128, 2
    for N, tile, chain in ((131072 + 1024 + 2, 0, [dspfx.Gain(0.9), dspfx.BiQuad(), dspfx.Reverb(delay_samples=200, decay=0.5), dspfx.LowPass(0.3),
                                                  dspfx.Distort(2.0, dspfx.RECIP_SOFT_CLIP), dspfx.Envelope(2.0, 80.0)]),
                           (131072 + 2048, 256, [dspfx.BiQuad(), dspfx.Distort(3.0, dspfx.TANH), dspfx.Overdrive(4.0, 0.6, 0.9),
                                                 dspfx.Chebyshev(3.0, 1.5), dspfx.SignalGen(0.2, 500.0, dspfx.SIG_SINE), dspfx.Gain(0.5)])):
        def run(c0, n):
            eng = dspfx.Engine(n, B, channel_offset=c0, tile_channels=tile)
            eng.set_chain(chain)
            eng.kernels_ready()              # the default choice once the background compiler is done with the shape
            stage = [l for l in eng.describe().splitlines() if l.startswith("stage")][0]
            outs, mixes = [], []
            for b in range(blocks):
                x = torch_cuda.empty(B * n, device="cuda")
                eng.fill_noise(x, B, b * B, 0x5EED0001)
                y = torch_cuda.empty_like(x)
                m = torch_cuda.empty(B, device="cuda")
                eng.process(x, out=y, mix=m, n_frames=B)
                torch_cuda.cuda.synchronize()
                outs.append(dspfx.from_layout(y.cpu().numpy(), B, n, tile))
                mixes.append(m.cpu().numpy().astype(np.float64))
            return np.concatenate(outs), np.concatenate(mixes), stage
        # frame-major case: the default choice (a run-time specialised kernel); tiled case: specialisation switched
        # off, so the default interpreter choice (two channels per lane) is what runs
        if tile:
            monkeypatch.setenv("DSPFX_JIT", "0")
        big, big_mix, stage = run(0, N)
        assert ("jit_" in stage) if not tile else ("dyn_libm_f8_c2" in stage), stage
        half = 65536 + 512
        a, am, sa = run(0, half)
        b, bm, _ = run(half, N - half)
        # the halves are small engines: one channel per lane; from 16384 channels on they too get a run-time specialised
        # kernel unless that is switched off (66048 channels is past the time-sliced kernel's range for a 6-node chain)
        assert "_c2" not in sa.split(";")[0] and (("jit_" in sa and "time-sliced" not in sa) if not tile else "jit_" not in sa), sa
        assert np.array_equal(big.view(np.uint32), np.concatenate([a, b], axis=1).view(np.uint32)), (N, tile)
        assert np.allclose(big_mix, am + bm, rtol=1e-5, atol=1e-2)


def test_runtime_specialised_kernels_match_interpreter_and_oracle(dspfx, torch_cuda, monkeypatch):
    """DSPFX_JIT=1 forces the hiprtc-instantiated `chain_kernel<F, CPL, SigList<...>>` at test sizes (large engines
    use it by themselves): every node kind on its own and seeded random chains must give the interpreter's bits and
    stay within the oracle bars; ragged channel counts, both layouts, mix bus, stage splits."""
    N, B, blocks = 128 * 3 + 34, 128, 3
    x, side = noise_block(N, B * blocks), noise_block(N, B * blocks, seed=9)
    exact, libm = _every_node(dspfx)
    nodes = exact + libm
    for node in nodes:
        bar = 1 if any(node is e for e in exact) else LIBM_COMPOSITE_ULP
        monkeypatch.setenv("DSPFX_JIT", "0")
        base, base_mix = run_gpu(dspfx, torch_cuda, [node], x, side=side, want_mix=True)
        monkeypatch.setenv("DSPFX_JIT", "1")
        eng = dspfx.Engine(N, B)
        eng.set_chain([node])
        assert "jit_" in eng.describe(), eng.describe()
        got, mix = run_gpu(dspfx, torch_cuda, [node], x, side=side, want_mix=True)
        assert np.array_equal(got.view(np.uint32), base.view(np.uint32)), (node.kind, node.mode)
        assert np.allclose(mix, base_mix, rtol=1e-5, atol=1e-3)
        assert ulp_diff(got, run_oracle([node], x, 3, side)).max() <= bar, (node.kind, node.mode)
    # a generator whose block ends inside a call (100-frame blocks): the specialised kernel closes the block too
    monkeypatch.setenv("DSPFX_JIT", "1")
    for mode in (dspfx.SIG_TRIANGLE, dspfx.SIG_SQUARE, dspfx.SIG_CONSTANT):
        chain = [dspfx.SignalGen(0.9, 777.0, mode), dspfx.Gain(0.5)]
        got = run_gpu(dspfx, torch_cuda, chain, noise_block(64, 600), block=100)
        ref = O.run_channels([n.oracle_desc() for n in chain], np.zeros((600, 1), F), block=100)
        assert np.array_equal(got[:, :1], ref), mode
    rng = np.random.default_rng(4242)
    for case in range(8):
        chain = [_random_exact_node(dspfx, rng) for _ in range(int(rng.integers(2, 13)))]
        tile = int(rng.choice([0, 64]))
        n_ch = 448 if tile else int(rng.choice([100, 129, 418]))
        lf = int(rng.choice([0, 1, 3]))
        xs, ss = noise_block(n_ch, B * 4, seed=case), noise_block(n_ch, B * 4, seed=case + 50)
        monkeypatch.setenv("DSPFX_JIT", "1")
        got = run_gpu(dspfx, torch_cuda, chain, xs, link_flags=lf, side=ss, tile=tile, block=int(rng.choice([128, 256])))
        monkeypatch.setenv("DSPFX_JIT", "0")
        base = run_gpu(dspfx, torch_cuda, chain, xs, link_flags=lf, side=ss, tile=tile)
        ok = np.isfinite(base)
        assert np.array_equal(got[ok].view(np.uint32), base[ok].view(np.uint32)), (case, [(n.kind, n.mode) for n in chain])
        ref = run_oracle(chain, xs, lf, ss)
        assert ulp_diff(got[ok], ref[ok]).max() <= 1, case
    monkeypatch.delenv("DSPFX_JIT", raising=False)


@pytest.mark.parametrize("lf", [0, 1, 3])
def test_long_chain_runs_as_one_generated_kernel(dspfx, torch_cuda, monkeypatch, lf):
    """A fusable run of 9..16 nodes on a large engine (DSPFX_JIT=1 stands in for > 131072 channels) is ONE generated
    kernel (the chain as a graph, hops per the engine's link flags) instead of two chain launches: same bits as the
    two-launch form, within the oracle's bar; a run with an Add / Mix (side input from memory) is cut as before; once
    control ports are used the engine goes back to chain launches."""
    torch = torch_cuda
    E = dspfx
    N, B, blocks = 192, 128, 4
    chain = [E.Gain(0.9), E.BiQuad(1.0, -1.2, 0.5, 0.3, 0.2, 0.1), E.Distort(3.0, E.SOFT_CLIP), E.LowPass(0.3),
             E.Reverb(delay_samples=256, decay=0.4), E.HighPass(0.2), E.Gain(1.1), E.Distort(2.0, E.HARD_CLIP),
             E.BiQuad(1.0, -0.5, 0.2, 0.4, 0.1, 0.0), E.Envelope(4.0, 100.0), E.Reverb(delay_samples=160, decay=0.3), E.Gain(0.7)]
    x = noise_block(N, B * blocks, seed=0x5EED000A)
    monkeypatch.setenv("DSPFX_JIT", "0")
    two = run_gpu(E, torch, chain, x, link_flags=lf)
    monkeypatch.setenv("DSPFX_JIT", "1")
    eng = E.Engine(N, B, link_flags=lf)
    eng.set_chain(chain)
    d = eng.describe()
    assert d.count("fused kernel") == 1 and "jit_graph" in d, d
    one, mix = run_gpu(E, torch, chain, x, link_flags=lf, want_mix=True)
    assert np.array_equal(one.view(np.uint32), two.view(np.uint32))
    assert np.allclose(mix, one.astype(np.float64).sum(axis=1), rtol=1e-5, atol=1e-3)
    assert ulp_diff(one, run_oracle(chain, x, lf)).max() <= 1
    # control ports: evaluated by the chain kernels -> the engine re-plans into chain launches, state carried over
    ctl_sig = noise_block(N, B * blocks, seed=0x5EED000B)
    ref_eng = E.Engine(N, B, link_flags=lf)
    monkeypatch.setenv("DSPFX_JIT", "0")
    ref_eng.set_chain(chain)
    for k in range(blocks):
        dx = torch.from_numpy(x[k * B:(k + 1) * B]).cuda()
        dc = torch.from_numpy(ctl_sig[k * B:(k + 1) * B]).cuda()
        ctl = {(0, 0): dc} if k >= 2 else None
        ya = eng.process(dx, out=torch.empty_like(dx), n_frames=B, ctl=ctl)
        yb = ref_eng.process(dx, out=torch.empty_like(dx), n_frames=B, ctl=ctl)
        torch.cuda.synchronize()
        assert torch.equal(ya.view(torch.int32), yb.view(torch.int32)), k
    assert eng.describe().count("fused kernel") == 2, eng.describe()
    eng.close()
    ref_eng.close()
    # an Add in the run reads the side input from memory: two chain launches as before
    monkeypatch.setenv("DSPFX_JIT", "1")
    eng = E.Engine(N, B, link_flags=lf)
    eng.set_chain(chain[:6] + [E.Add()] + chain[6:])
    assert eng.describe().count("fused kernel") == 2 and "jit_graph" not in eng.describe(), eng.describe()
    eng.close()


# ---------------------------------------------------------------- Reverb: every slider store swaps in a new zero ring

def _run_with_stores(dspfx, torch, eng, x, actions, block, tile=0):
    """x [frames][N] block by block through `eng`; actions = {block index: callable(eng)} run BEFORE that block."""
    nf, N = x.shape
    y = np.empty_like(x)
    for k, f0 in enumerate(range(0, nf, block)):
        n = min(block, nf - f0)
        if k in actions:
            actions[k](eng)
        dx = torch.from_numpy(dspfx.to_layout(x[f0:f0 + n], tile)).cuda()
        dy = torch.empty_like(dx)
        eng.process(dx, out=dy, n_frames=n)
        torch.cuda.synchronize()
        y[f0:f0 + n] = dspfx.from_layout(dy.cpu().numpy(), n, N, tile)
    return y


def _oracle_with_stores(chain, x, actions, block, link_flags=3):
    """The same through per-channel oracle nodes; actions = {block index: callable(list of nodes of one channel)}."""
    descs = [n.oracle_desc() for n in chain]
    out = np.empty_like(x)
    for c in range(x.shape[1]):
        nodes = [O.node_from_desc(d) for d in descs]
        for k, f0 in enumerate(range(0, x.shape[0], block)):
            if k in actions:
                actions[k](nodes)
            out[f0:f0 + block, c] = O.chain_run(nodes, x[f0:f0 + block, c], link_flags, block=min(block, 128))
    return out


@pytest.mark.parametrize("N,tile,variant,block,D", [
    (192, 0, "static=0,f=8,cpl=1", 128, 203),        # the interpreter; 203 - 128 = 75 cleared rows end inside a chunk
    (256, 0, "static=0,f=8,cpl=2", 128, 203),        # ... two channels per lane
    (192, 0, "static=1,f=8,cpl=1", 128, 203),        # the statically specialised chain3 / chain5 kernels
    (512, 256, "static=1,f=8,cpl=2", 128, 128),      # D == block: every row of the block is cleared, none of the next
    (256, 0, "static=1,f=16,cpl=1", 100, 1000),      # blocks of 100 frames: the one-frame remainder chunks
    (512, 256, "static=1,f=8,cpl=4", 256, 300),      # 256-frame blocks over a 300-sample ring
    (64 * 9, 64, "ts=1", 128, 203),                  # the time-sliced kernel (four slices: 75 ends inside slice 2)
    (64 * 5 + 17, 0, "ts=1", 128, 300),              # ... and its guarded form for the channels left over
    (131, 0, None, 128, 203),                        # defaults, ragged: the guarded interpreter tail
])
@pytest.mark.parametrize("which", ["chain3", "chain5"])
def test_reverb_slider_store_swaps_in_a_zero_ring(dspfx, torch_cuda, monkeypatch, which, N, tile, variant, block, D):
    """reverb.rs:19 + dsp-stuff-derive/src/lib.rs:560-568: ANY slider store on a Reverb node -- `decay` included -- runs
    refresh_seconds, i.e. a new zero-filled ring (reverb.rs:55-71).  The engine does it without touching the ring: the next D
    frames read their taps as +0.0 (SlotArgs::zero_rows).  Against the oracle with the same stores at the same blocks, in
    every kernel family; the second store lands while the first clear is still counting down."""
    if variant:
        monkeypatch.setenv("DSPFX_VARIANT", variant)
    chain = chain3(dspfx, D) if which == "chain3" else chain5(dspfx, D)
    ridx = 2
    nblocks = max(8, 3 * D // block + 4)
    x = noise_block(N, block * nblocks)
    k1 = D // block + 2                                # the ring is full of echoes by then
    k2 = k1 + 1 if D > block else k1 + 2               # second store: inside the first clear's countdown when D > block
    k3 = k2 + D // block + 2
    eng = dspfx.Engine(N, block, link_flags=3, tile_channels=tile)
    eng.set_chain(chain)
    if variant and variant.startswith("ts"):
        assert "time-sliced" in eng.describe(), eng.describe()
    gpu_actions = {k1: lambda e: e.set_param(ridx, 0, 0.7), k2: lambda e: e.set_param(ridx, 0, 0.3),
                   k3: lambda e: e.set_delay_len(ridx, D)}          # an unchanged D: the same O(1) path
    orc_actions = {k1: lambda ns: ns[ridx].set_param(0, 0.7), k2: lambda ns: ns[ridx].set_param(0, 0.3),
                   k3: lambda ns: ns[ridx].set_delay_len(D)}
    y = _run_with_stores(dspfx, torch_cuda, eng, x, gpu_actions, block, tile)
    ref = _oracle_with_stores(chain, x, orc_actions, block)
    assert ulp_diff(y, ref).max() <= 1, (ulp_diff(y, ref).max(), np.argwhere(ulp_diff(y, ref) > 1)[:4])
    # ... and the tail really was cut: ignoring the hook (the round-3 behaviour) gives something else
    orc_plain = {k1: lambda ns: ns[ridx].L.orc_node_init_param(ns[ridx].h, 0, 0.7), k2: lambda ns: ns[ridx].L.orc_node_init_param(ns[ridx].h, 0, 0.3)}
    assert ulp_diff(y, _oracle_with_stores(chain, x, orc_plain, block)).max() > 1000
    # the exported ring: what a block would read -- zeros for the rows from before the last clear, samples for the rest
    st = eng.state_export(ridx).view(np.float32).reshape(D, N)
    eng.set_delay_len(ridx, D)
    assert not eng.state_export(ridx).any()
    assert st.any()
    eng.close()


def test_reverb_fresh_node_and_seconds_slider_resize_the_ring(dspfx, torch_cuda):
    """A node fresh from the menu keeps make_buffer()'s 128-sample ring until its first slider change, which sizes the ring
    from the seconds slider (reverb.rs:44-52, 55-71); a seconds store re-sizes it again; with the page-rounded reading of
    rivulet's capacity (mode bit 0) 0.01 s is 1024 samples.  A changed length re-allocates the ring at that block boundary."""
    N, B = 96, 128
    for page_round in (False, True):
        chain = [dspfx.Gain(0.9), dspfx.Reverb(page_round=page_round), dspfx.LowPass(0.3)]
        assert chain[1].delay_len == (1024 if page_round else 128) and list(chain[1].params) == [0.5, 0.5]   # make_buffer(), either reading
        x = noise_block(N, B * 24)
        eng = dspfx.Engine(N, B, link_flags=3)
        eng.set_chain(chain)
        gpu = {3: lambda e: e.set_param(1, 1, 0.01), 14: lambda e: e.set_param(1, 0, 0.4), 19: lambda e: e.set_param(1, 1, 0.004)}
        orc = {3: lambda ns: ns[1].set_param(1, 0.01), 14: lambda ns: ns[1].set_param(0, 0.4), 19: lambda ns: ns[1].set_param(1, 0.004)}
        y = _run_with_stores(dspfx, torch_cuda, eng, x, gpu, B)
        ref = _oracle_with_stores(chain, x, orc, B)
        assert ulp_diff(y, ref).max() <= 1
        assert len(eng.state_export(1)) == 4 * N * (O.delay_len(0.004, page_round))
        eng.close()
    # the default half second: 128 -> 24000 samples at the first touch of `decay`
    eng = dspfx.Engine(64, B, link_flags=0)
    eng.set_chain([dspfx.Reverb()])
    assert len(eng.state_export(0)) == 4 * 64 * 128
    eng.set_param(0, 0, 0.5)
    assert len(eng.state_export(0)) == 4 * 64 * 24000
    eng.close()
    # a node given an explicit ring and no seconds slider keeps its length through decay stores; a STORED seconds of 0.0 is a
    # value like any other: max((0 * 48000) as usize, 128) = 128 samples (reverb.rs:58), and it stays the slider for later hooks
    eng = dspfx.Engine(64, B, link_flags=0)
    eng.set_chain([dspfx.Reverb(delay_samples=300)])
    eng.set_param(0, 0, 0.4)
    assert len(eng.state_export(0)) == 4 * 64 * 300
    eng.set_param(0, 1, 0.0)
    assert len(eng.state_export(0)) == 4 * 64 * 128
    eng.set_param(0, 0, 0.3)
    assert len(eng.state_export(0)) == 4 * 64 * 128
    eng.close()


@pytest.mark.parametrize("N,tile,variant,block", [
    (192, 0, "static=0,f=8,cpl=1", 128),             # the interpreter
    (512, 256, "static=1,f=8,cpl=2", 128),           # the statically specialised kernels, tiled
    (256, 0, "static=1,f=16,cpl=1", 100),            # ragged blocks
    (64 * 9, 64, "ts=1", 128),                       # the time-sliced kernel (group pointers from the host)
    (64 * 5 + 17, 0, "ts=1", 128),                   # ... and its guarded form
    (131, 0, None, 128),                             # defaults
])
def test_reverb_length_changes_reuse_the_rings_groups(dspfx, torch_cuda, monkeypatch, N, tile, variant, block):
    """A ring is the first ceil(D / 128) of the node's 128-row groups (include/dspfx.h, dspfx_set_param): growing appends
    groups (never zeroed: the lazy clear masks whatever they hold), shrinking keeps the surplus, growing again re-uses
    groups still FULL of the longer ring's old echoes -- which must read as the zeros of refresh_seconds' new ring
    (reverb.rs:55-71).  203 -> 1000 -> 300 -> 700 (inside the first clear's countdown) -> 2000 samples against the oracle,
    in every kernel family; nothing waits for the device and the capacity only ever grows."""
    if variant:
        monkeypatch.setenv("DSPFX_VARIANT", variant)
    chain = chain5(dspfx, 203)
    ridx = 2
    lens = [1000, 300, 700, 2000]
    at, k = {}, 4
    for i, D in enumerate(lens):
        at[k] = D
        k += (2 if i == 1 else D // block + 3)       # the third change lands while the second one's clear is still counting
    nblocks = k + 4
    x = noise_block(N, block * nblocks)
    eng = dspfx.Engine(N, block, link_flags=3, tile_channels=tile)
    eng.set_chain(chain)
    gpu = {kk: (lambda e, D=D: e.set_delay_len(ridx, D)) for kk, D in at.items()}
    orc = {kk: (lambda ns, D=D: ns[ridx].set_delay_len(D)) for kk, D in at.items()}
    y = _run_with_stores(dspfx, torch_cuda, eng, x, gpu, block, tile)
    ref = _oracle_with_stores(chain, x, orc, block)
    assert ulp_diff(y, ref).max() <= 1, (ulp_diff(y, ref).max(), np.argwhere(ulp_diff(y, ref) > 1)[:4])
    assert "2000 samples in 16 of 16 groups" in eng.describe(), eng.describe()
    eng.set_delay_len(ridx, 128)
    assert "128 samples in 1 of 16 groups" in eng.describe(), eng.describe()
    eng.ring_trim()
    assert "128 samples in 1 of 1 groups" in eng.describe(), eng.describe()
    eng.close()


def test_menu_fresh_reverb_ring_is_reserved_only_when_cheap_and_quietly(dspfx, torch_cuda, monkeypatch):
    """dspfx_chain_set used to reserve the whole seconds-slider ring of every menu-fresh REVERB node -- 23.5 GiB
    at 262 144 channels, 94 GiB at 2^20 -- for a slider that may never move.  Now: only when the missing groups take at most 1/16
    of the device's free memory (DSPFX_MENU_RING_RESERVE=1: whenever they fit; =0: never), never with a trace in dspfx_last_error;
    the first slider store allocates on the storing thread like the reference's GUI thread (reverb.rs:55-71) and
    dspfx_reserve_delay_len is the explicit hint."""
    B = 128
    small = dspfx.Engine(4096, B, link_flags=0)
    small.set_chain([dspfx.Reverb()])                                     # 187 more groups of 2 MiB: cheap
    assert "128 samples in 1 of 1 groups" in small.describe() and "(+187 reserved)" in small.describe(), small.describe()
    small.close()
    big = dspfx.Engine(1 << 18, B, link_flags=0)
    big.set_chain([dspfx.Reverb()])                                       # 187 groups of 128 MiB = 23.4 GiB: not taken behind the host's back
    assert "(+0 reserved)" in big.describe(), big.describe()
    assert big.L.dspfx_last_error(big.h).decode() == "", big.L.dspfx_last_error(big.h)
    x = torch_cuda.zeros((B, 1 << 18), dtype=torch_cuda.float32, device="cuda")
    x[0, :] = 1.0
    y = torch_cuda.empty_like(x)
    big.process(x, out=y, n_frames=B)
    big.set_param(0, 0, 0.25)                                             # the first touch: the storing thread allocates, the swap is O(1)
    assert "24000 samples in 188 of 188 groups" in big.describe(), big.describe()
    big.reserve_delay_len(0, 48000)                                       # the explicit hint still works at any size
    assert "(+187 reserved)" in big.describe(), big.describe()
    big.ring_trim()
    assert "(+0 reserved)" in big.describe()
    monkeypatch.setenv("DSPFX_MENU_RING_RESERVE", "1")
    big.set_chain([dspfx.Reverb()])
    assert "(+187 reserved)" in big.describe(), big.describe()
    big.close()
    monkeypatch.setenv("DSPFX_MENU_RING_RESERVE", "0")
    small = dspfx.Engine(4096, B, link_flags=0)
    small.set_chain([dspfx.Reverb()])
    assert "(+0 reserved)" in small.describe(), small.describe()
    small.close()


def test_reverb_seconds_store_that_cannot_be_had_leaves_the_node_alone(dspfx, torch_cuda):
    """a seconds store whose ring does not fit the device fails on the STORING thread with DSPFX_ERR_OOM, before
    anything is queued: the node keeps ring, length and slider, and goes on processing; a value outside the slider's 0..=1
    (reverb.rs:34-37) is DSPFX_ERR_INVALID.  4 194 304 channels: one 128-row group is 2 GiB, one second would be 750 GiB."""
    torch = torch_cuda
    N, B = 1 << 22, 128
    eng = dspfx.Engine(N, B, link_flags=0)
    eng.set_chain([dspfx.Reverb()])                    # menu-fresh: 128 samples under a 0.5 s slider; 376 GiB cannot be reserved
    assert "128 samples in 1 of 1 groups" in eng.describe() and "(+0 reserved)" in eng.describe(), eng.describe()
    x = torch.zeros((B, N), dtype=torch.float32, device="cuda")
    x[0, :] = 1.0
    y = torch.empty_like(x)
    eng.process(x, out=y, n_frames=B)
    with pytest.raises(dspfx.DspfxError) as ei:
        eng.set_param(0, 1, 1.0)
    assert ei.value.status == -4 and "no room" in str(ei.value), ei.value
    for bad in (1.5, -0.1, float("nan")):
        with pytest.raises(dspfx.DspfxError) as ei:
            eng.set_param(0, 1, bad)
        assert ei.value.status == -1
    assert eng.param_log() == []
    x.zero_()
    eng.process(x, out=y, n_frames=B)                  # the 128-sample ring is still there: the impulse comes back halved
    torch.cuda.synchronize()
    assert float(y[0].min()) == 0.5 and float(y[0].max()) == 0.5 and float(y[1:].abs().max()) == 0.0
    with pytest.raises(dspfx.DspfxError):              # `decay` on this node asks for the half-second ring too (lib.rs:560-568)
        eng.set_param(0, 0, 0.25)
    eng.set_param(0, 1, 0.004)                         # a ring that fits: 192 samples, two groups
    assert "192 samples in 2 of 2 groups" in eng.describe(), eng.describe()
    eng.close()


def test_reverb_store_inside_a_generated_graph_kernel(dspfx, torch_cuda):
    """The whole-graph kernel shares the ring code: a decay store on a Reverb node of a fused DAG cuts its tail too."""
    N, B, D = 256, 128, 203
    nodes = [dspfx.Gain(0.8), dspfx.Reverb(delay_samples=D, decay=0.6), dspfx.HighPass(0.2), dspfx.Add()]
    links = [(dspfx.GRAPH_INPUT, 0, dspfx.PORT_MAIN), (0, 1, dspfx.PORT_MAIN), (0, 2, dspfx.PORT_MAIN), (1, 3, dspfx.PORT_MAIN),
             (2, 3, dspfx.PORT_SIDE), (3, 4, dspfx.PORT_MAIN)]
    x = noise_block(N, B * 10)
    eng = dspfx.Engine(N, B, link_flags=3)
    eng.set_graph(nodes, links)
    y = _run_with_stores(dspfx, torch_cuda, eng, x, {4: lambda e: e.set_param(1, 0, 0.2), 5: lambda e: e.reset()}, B)
    eng.close()
    ref = np.empty_like(x)
    for c in range(0, N, 37):
        ns = [O.node_from_desc(n.oracle_desc()) for n in nodes]
        for k in range(10):
            if k == 4:
                ns[1].set_param(0, 0.2)
            if k == 5:
                ns = [O.node_from_desc(dict(n.oracle_desc(), params=[0.2, 0.0] if i == 1 else n.oracle_desc()["params"])) for i, n in enumerate(nodes)]
            seg = x[k * B:(k + 1) * B, c]
            hop = lambda v: (np.zeros(B, F) + v) / O.link_divisor(1)
            g = ns[0].process(hop(seg))
            r = ns[1].process(hop(g))
            h = ns[2].process(hop(g))
            a = ns[3].process(hop(r), hop(h))
            ref[k * B:(k + 1) * B, c] = hop(a)
        assert ulp_diff(y[:, c], ref[:, c]).max() <= 1, c


# ---------------------------------------------------------------- f3: a WAV impulse response all the way to the FIR kernels

def test_wav_impulse_responses_through_the_fir_kernels(dspfx, torch_cuda, tmp_path):
    """nodes/fir.rs:86-173 -> 179-225 end to end: a WAV file is decoded, its channels averaged, a file that is not 48 kHz
    resampled (ir.load_impulse_response / include/dspfx_ir.hpp), the taps reversed (Fir(...), fir.rs:163,168) and handed to
    dspfx_chain_set; the GPU's FIR kernels against the oracle fed the same taps -- bit for bit where the data is exact in f32
    (16-bit taps k / 32768, small integer samples), within the stated 1e-6 relative RMS otherwise -- and the same file through
    the C++ loader and engine wrapper (tests/cpp/test_host --ir), bit for bit against the Python path."""
    import subprocess
    from dsp_stuff_amd import ir
    from test_config_ir_cpu import _wav
    import test_cpp_host
    test_cpp_host._build()
    rng = np.random.default_rng(12)
    N, blocks = 64, 6
    xn = O.noise(0x5EED0004, np.arange(N), np.arange(128 * blocks))
    cases = {
        "mono16_exact": (1, 16, 48000, rng.integers(-4, 5, (200, 1)) / 32768.0, np.round(xn * 8).astype(F)),
        "stereo_float": (3, 32, 48000, rng.uniform(-0.5, 0.5, (300, 2)).astype(np.float32).astype(np.float64), xn),
        "mono16_44k1": (1, 16, 44100, np.round(rng.uniform(-0.9, 0.9, (150, 1)) * 32768) / 32768, xn),
    }
    for name, (tag, bits, rate, frames, x) in cases.items():
        p = str(tmp_path / (name + ".wav"))
        _wav(p, tag, bits, rate, frames)
        h = ir.load_impulse_response(p)
        assert len(h) == (len(frames) if rate == 48000 else len(ir.resample_dasp_sinc(frames.mean(axis=1), rate)))
        if name == "stereo_float":
            assert np.array_equal(h, (frames[:, 0] + frames[:, 1]) / 2.0)          # fir.rs:140-144
        chain = [dspfx.Fir(h)]
        y = run_gpu(dspfx, torch_cuda, chain, x, link_flags=0)
        ref = run_oracle(chain, x, 0)
        if name == "mono16_exact":
            assert np.array_equal(y.view(np.uint32), ref.view(np.uint32)), name
        else:
            assert fir_rel_rms(y, ref) < FIR_RMS_TOL, (name, fir_rel_rms(y, ref))
        if name != "mono16_exact":                                              # the C++ rig feeds the hashed noise itself
            r = subprocess.run([test_cpp_host.EXE, "--ir", p, str(blocks)], capture_output=True, text=True)
            assert r.returncode == 0, r.stdout + r.stderr
            lines = r.stdout.strip().splitlines()
            assert lines[0] == "taps %d" % len(h), lines[0]
            got = np.array([[float.fromhex(v) for v in l.split()] for l in lines[1:]], F)
            assert got.shape == (128 * blocks, 2)
            assert np.array_equal(got[:, 0].view(np.uint32), y[:, 0].view(np.uint32)) and np.array_equal(got[:, 1].view(np.uint32), y[:, 63].view(np.uint32)), name


def test_kernel_selection_by_engine_size(dspfx, torch_cuda):
    """Which kernel serves whole 128-frame blocks of the BASELINE chains at which size (plan.hip / jit.hip: the measured rules of
    profiles/r04_midn.txt), read from dspfx_describe: the time-sliced kernel holds one resident round (one channel per lane up to
    49152 channels for the 3-node chain, two from there); beyond it short chains in the tiled layout take the standard kernel
    at two channels per lane (at most one workgroup per CU up to 131072 channels), long chains and other layouts one channel per
    lane with 8-frame chunks; from 229376 channels on everything takes two channels per lane."""
    from chains import chain3
    expect = [
        ("chain3", 32768, 256, "time-sliced s3h_ts32_c1"), ("chain3", 49152, 256, "time-sliced s3h_ts32_c2"),
        ("chain3", 65536, 256, "time-sliced s3h_ts32_c2"), ("chain3", 81920, 256, "s3h_f8_c2"), ("chain3", 131072, 256, "s3h_f8_c2"),
        ("chain3", 81920, 0, "time-sliced s3h_ts32_c2"), ("chain3", 147456, 256, "s3h_f8_c1"), ("chain3", 229376, 256, "s3h_f8_c2"),
        ("chain5", 32768, 256, "time-sliced s5h_ts32_c1"), ("chain5", 65536, 256, "time-sliced s5h_ts32_c1"),
        ("chain5", 81920, 256, "s5h_f8_c1"), ("chain5", 131072, 256, "s5h_f8_c1"), ("chain5", 229376, 256, "s5h_f8_c2"),
    ]
    for which, N, tile, want in expect:
        eng = dspfx.Engine(N, 128, link_flags=3, tile_channels=tile)
        eng.set_chain(chain3(dspfx, 128) if which == "chain3" else chain5(dspfx, 128))
        stage = [l for l in eng.describe().splitlines() if l.startswith("stage")][-1]
        if "time-sliced" in want:
            assert want in stage, (which, N, tile, stage)
        else:
            assert ("fused kernel " + want) in stage and "time-sliced" not in stage, (which, N, tile, stage)
        eng.close()
