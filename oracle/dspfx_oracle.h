/*
 * dspfx_oracle.h -- CPU restatement of simmsb/dsp-stuff's per-block effect-node
 * evaluation loop.
 *
 * THIS IS TEST INFRASTRUCTURE, NOT PRODUCT CODE.  Only tests/, the smoke check in
 * __graft_entry__.py and bench.py's cpu_baseline leg may load it; it is the
 * checker, never the thing measured or shipped.  The product (libdspfx.so, HIP)
 * never links or calls anything in this directory.
 *
 * PARITY UNPINNED: the reference ships no tests, golden vectors or fixtures
 *  and its Rust-nightly toolchain is absent from this
 * image, so the restatement cannot be checked against outputs of the reference
 * itself.  It is pinned only by known-answer tests derived from the cited source
 * lines (tests/test_oracle_kat.py) and by an independent numpy-float32 model
 * (oracle/numpy_model.py).
 * To pin it, ONE command in a checkout of the reference (oracle/pin_kit/README.md):
 *   cp oracle/pin_kit/golden_dump.rs <reference>/dsp-stuff/src/  (+ `mod golden_dump;`
 *   and the early return in main.rs shown there), then
 *   DSPFX_GOLDEN_DUMP=oracle/pin_kit/cases.json DSPFX_GOLDEN_OUT=pin_out.json cargo run --release
 *   python tools/compare_pin.py pin_out.json
 * runs the REAL nodes over every golden vector of this repository through
 * Perform::perform and compares bit patterns with the parity tests' bars; it also
 * reports the two facts restated "as recalled" below (rivulet's granted view length,
 * DirectForm1::run's operation order).
 *
 * Third-party arithmetic that is NOT under /root/reference and is restated from
 * its published definition (Cargo.lock pins):
 *   - biquad 0.4.2  DirectForm1<f32>::{new,run,reset_state,update_coefficients}
 *   - rivulet @b2416e5 circular_buffer (FIFO semantics + initial zero fill only)
 *   - Rust std f32 math on linux-gnu == glibc libm (tanhf/sinf/atanf/expf)
 *   - Rust std VecDeque<f64> growth policy (for Fir's a/b slice split)
 *   - dasp_envelope 0.11.0 / dasp_peak 0.11.0 (Cargo.lock:1207-1246): Detector::{peak, set_attack_frames,
 *     set_release_frames, next} with the Peak<FullWave> rectifier, restated from the crate's published source
 *     AS RECALLED; not verifiable in this container.  The algorithm this restatement follows, written out so that
 *     it can be checked against the crate wherever its source is at hand (call sites: nodes/envelope.rs:45-51):
 *         calc_gain(n_frames)        = n_frames == 0 ? 0 : powf(E, -1 / n_frames)           (f32, E = core::f32::consts::E)
 *         Detector::next(frame x):     d    = full_wave(x) = x < 0 ? -x : x                  (dasp_peak::FullWave)
 *                                      gain = last_env < d ? attack_gain : release_gain
 *                                      env  = d + (last_env - d) * gain ;  last_env = env    (starts at equilibrium, 0)
 *     Closed forms that follow (tests/test_oracle_kat.py, KAT-11): a unit step from rest gives env[n] = 1 - g_a^(n+1);
 *     after the input returns to 0 the envelope decays as env[n] = env0 * g_r^(n+1); 0 frames => gain 0 => env = |x|.
 *   - dasp_signal 0.11.0 interpolate::Converter + dasp_interpolate 0.11.0 sinc::Sinc over ring_buffer::Fixed<[f64; 16]>
 *     (Cargo.lock; call site nodes/fir.rs:153-165: from_hz_to_hz(Sinc::new(Fixed::from([0.0; 16])), rate, 48000)),
 *     restated AS RECALLED in dsp-stuff_amd/ir.py and include/dspfx_ir.hpp (bit-identical to each other):
 *         depth = ring length / 2 = 8; a new source frame is pushed at the back of the ring, idx climbs 0 -> depth
 *         Converter: interpolation_value += source_hz / target_hz per output; every whole unit pulls one source frame
 *         Sinc::interpolate(x = interpolation_value in [0, 1)):
 *             v = sum over n in [0, max_depth) of  w(x + n) * ring[idx - n]  +  w(1 - x + n) * ring[idx + 1 + n]
 *             w(a) = sinc(pi a) * (0.5 + 0.5 cos(pi a / depth)),  sinc(0) = 1           (Hann-windowed sinc, 16 taps)
 *             max_depth = depth, clipped where idx +- depth leaves the ring; the ring indexes MODULO its length, so at
 *             idx = depth the outermost right tap (n = 7, ring[16]) reads ring[0], the oldest frame (weight <= 4e-4)
 *     Closed forms that follow (tests/test_oracle_kat.py, KAT-12): the output is the source delayed by `depth` frames; at
 *     integer rate ratios (1:1, 2:1) every output lands on a source frame and is EXACT (w(n) = 0 for n != 0); at
 *     44.1 -> 48 kHz a sinusoid below 10 kHz is interpolated to within 2.5e-3 of sin(2 pi f t) (measured 1.2e-3: the
 *     16-tap Hann window's passband ripple), and the output equals the direct evaluation of the formula above.
 *
 *   - Rust std `impl Sum for f64` (nodes/fir.rs:143, 206, 211): the reference pins `channel = "nightly"` with no date
 *     (rust-toolchain.toml:3).  Up to Rust 1.82 the fold starts from +0.0; since 1.83 it starts from -0.0, the true additive
 *     identity.  The two differ in ONE case only: a sum whose every term is -0.0 (or an empty one) gives -0.0 on a new toolchain
 *     and +0.0 on an old one.  This restatement (orc_fir_*: accumulators from +0.0), oracle/numpy_model.py and the exact f64 GPU
 *     kernel all take the OLD convention (which toolchain the reference is built with is not recorded anywhere in its tree); "bit-identical to the oracle"
 *     for FIR is a statement about that convention.  Nothing but the sign of such a zero depends on it (no bar does).
 *
 * Build: gcc -O2 -ffp-contract=off -fno-fast-math  (Rust never contracts a*b+c).
 * All citations are relative to /root/reference/.
 */
#ifndef DSPFX_ORACLE_H
#define DSPFX_ORACLE_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* dsp-stuff/src/node.rs:257 */
#define ORC_BUF_SIZE 128

/* Node kinds.  Numeric values are shared with include/dspfx.h (dspfx_kind). */
enum {
    ORC_GAIN = 0,      /* nodes/gain.rs */
    ORC_BIQUAD = 1,    /* nodes/biquad.rs */
    ORC_LOW_PASS = 2,  /* nodes/low_pass.rs */
    ORC_HIGH_PASS = 3, /* nodes/high_pass.rs */
    ORC_REVERB = 4,    /* nodes/reverb.rs (feedback delay line) */
    ORC_DISTORT = 5,   /* nodes/distort.rs */
    ORC_OVERDRIVE = 6, /* nodes/overdrive.rs */
    ORC_CHEBYSHEV = 7, /* nodes/chebyshev.rs */
    ORC_FIR = 8,       /* nodes/fir.rs */
    ORC_ADD = 9,       /* nodes/add.rs */
    ORC_MIX = 10,      /* nodes/mix.rs */
    ORC_SIGNAL_GEN = 11, /* nodes/signal_gen.rs: a SOURCE (no "in" port); its input is ignored */
    ORC_ENVELOPE = 12,   /* nodes/envelope.rs over dasp_envelope 0.11.0 Detector<f32, Peak<FullWave>> */
    ORC_N_KINDS = 13
};

/* nodes/distort.rs:18-28, in declaration order (repr(u8)). */
enum {
    ORC_DIST_HARD_CLIP = 0,
    ORC_DIST_SOFT_CLIP = 1,
    ORC_DIST_TANH = 2,
    ORC_DIST_RECIP_SOFT_CLIP = 3,
    ORC_DIST_FUZZ = 4,
    ORC_DIST_SIN = 5,
    ORC_DIST_ATAN = 6,
    ORC_DIST_SQUARE = 7,
    ORC_DIST_CHEBYSHEV4 = 8
};

/* nodes/signal_gen.rs:17-22 Mode */
enum { ORC_SIG_SINE = 0, ORC_SIG_TRIANGLE = 1, ORC_SIG_SQUARE = 2, ORC_SIG_CONSTANT = 3 };

/* nodes/fir.rs Mode */
enum { ORC_FIR_BALANCED = 0, ORC_FIR_AVERAGE = 1 };

/*
 * EVERY path of the cited nodes that touches DSP state outside `process`, and where this file restates it
 * (the Reverb hook on `decay` had slipped through both restatements):
 *
 *   reference path                                         | what it does to state                      | restated by
 *   -------------------------------------------------------+--------------------------------------------+------------------------------
 *   render(): any widget of the node changed               | runs after_settings_change (below)         | orc_node_set_param (every slot),
 *     dsp-stuff-derive/src/lib.rs:487-497,560-568,570-578   |                                            | (select widgets set `changed` too,
 *                                                          |                                            | lib.rs:521-525, but no node with a
 *                                                          |                                            | hook has one: orc_node_set_mode
 *                                                          |                                            | is the plain store)
 *   BiQuad::regenerate_filter  biquad.rs:15, 62-76         | coeffs / a0; reset_state(): x1,x2,y1,y2 = 0 | biquad_regenerate
 *   Reverb::refresh_seconds    reverb.rs:19, 55-71         | NEW ring of max(seconds*48000,128) zeros   | orc_node_after_settings_change ->
 *                                                          |                                            | orc_reverb_set_len
 *   restore(): after the field setters                     | runs after_settings_change ONCE            | orc_node_after_settings_change
 *     dsp-stuff-derive/src/lib.rs:319-337                   | (biquad filter rebuilt, reverb ring sized)  | (oracle.py Node(restored=True))
 *   new() (menu): field defaults only, NO hook             | BiQuad::initial_filter biquad.rs:48-60;    | orc_node_new
 *     dsp-stuff-derive/src/lib.rs:196-210                   | make_buffer(): 128 zeros reverb.rs:44-52   |
 *   `<slider>_input` latch  lib.rs:148                      | slider atomic <- first mapped sample of    | orc_slider_input (*atomic = out[0])
 *                                                          | the block (as_input sliders only)          |
 *   Fir: custom_render, no hook; load_file fir.rs:153-171  | replaces `taps` ONLY, `state` kept         | orc_fir_set_taps
 *   no other node on the path has a hook, a Mutex'd state or a custom render (grep after_settings_change|custom_render
 *   nodes/: biquad, reverb, fir only); low/high-pass `z`, signal_gen `clock`, envelope `detector` change in process() only.
 *
 * One node instance for ONE mono channel: parameters (the reference's slider
 * atomics) + DSP state.  Parameter slots by kind (same slot numbering as
 * dspfx_node_desc.params in include/dspfx.h):
 *   GAIN       p[0]=level
 *   BIQUAD     p[0..5]=a0,a1,a2,b0,b1,b2 (raw sliders, normalised by a0)
 *   LOW_PASS   p[0]=ratio      HIGH_PASS p[0]=ratio
 *   REVERB     p[0]=decay, p[1]=seconds (0 = not given), ip[0]=D (delay length in samples, explicit), mode bit 0 = the
 *              page-rounded reading of seconds -> samples when a refresh derives D from the seconds slider
 *   DISTORT    p[0]=level, mode
 *   OVERDRIVE  p[0]=boost, p[1]=drive, p[2]=level
 *   CHEBYSHEV  p[0]=level_pos, p[1]=level_neg
 *   FIR        taps (time-reversed, as stored by fir.rs:163,168), mode
 *   ADD        -              MIX p[0]=ratio
 *   SIGNAL_GEN p[0]=amplitude, p[1]=frequency, mode; state: clock
 *   ENVELOPE   p[0]=attack, p[1]=release (both in frames, sliders 0..=1000); state: env
 */
typedef struct orc_node {
    int kind;
    int mode;
    float p[8];
    uint32_t ip[2];
    int seconds_given; /* REVERB: the seconds slider is known (initialised > 0, or stored since -- a stored 0.0 counts: reverb.rs:58 gives 128) */

    /* biquad: normalised coefficients + DirectForm1 state */
    float bq_a1, bq_a2, bq_b0, bq_b1, bq_b2;
    float bq_x1, bq_x2, bq_y1, bq_y2;
    /* one-pole z (low_pass.rs:23 / high_pass.rs:23) */
    float z;
    /* signal generator phase (signal_gen.rs:52) */
    float clock;
    /* envelope follower: Detector::last_env_frame (envelope.rs:25) */
    float env;
    /* reverb ring: exactly D samples, FIFO (read oldest, append newest) */
    float *ring;
    uint32_t ring_len, ring_pos;
    /* fir: taps (reversed) + VecDeque<f64> emulation */
    double *taps;
    uint32_t n_taps;
    double *dq;
    uint32_t dq_cap, dq_head, dq_len;
} orc_node;

/* ---- node lifecycle ---------------------------------------------------- */
/* Create with the reference's defaults for `kind` (derive `default=`). */
orc_node *orc_node_new(int kind);
void orc_node_free(orc_node *n);
/* A slider change made in the GUI: store slot `idx`, then the node's after_settings_change hook (see the table below):
 * BIQUAD renormalises + RESETS its state; REVERB -- for ANY slot, decay included -- swaps in a new zero ring. */
void orc_node_set_param(orc_node *n, int idx, float v);
/* The plain field store of new() / restore(): no hook. */
void orc_node_init_param(orc_node *n, int idx, float v);
/* The hook alone (what restore() runs once after setting every field). */
void orc_node_after_settings_change(orc_node *n);
void orc_node_set_mode(orc_node *n, int mode);
/* reverb.rs:55-71 with D explicit: fresh zero-filled ring of D samples. */
void orc_reverb_set_len(orc_node *n, uint32_t d);
/* make_buffer() under the reading the node's mode bit 0 selects: 128 samples, or 1024 (= orc_delay_len(0, 1)) page-rounded.
 * orc_node_new leaves the unrounded 128; a fresh node whose mode says page-rounded calls this once after orc_node_set_mode. */
void orc_reverb_make_buffer(orc_node *n);
/* reverb.rs:58 helper: max((seconds*48000f32) as usize, 128); page_round!=0
 * additionally rounds up to whole 4 KiB pages (1024 f32) -- the two candidate
 * readings of rivulet's capacity rounding. */
uint32_t orc_delay_len(float seconds, int page_round);
/* fir.rs:61-62,163,168: taps are given ALREADY time-reversed (as stored). */
void orc_fir_set_taps(orc_node *n, const double *taps_reversed, uint32_t n_taps);
/* Zero all DSP state (fresh node). */
void orc_node_reset(orc_node *n);

/* ---- the hot path ------------------------------------------------------ */
/* node.rs:162-194: out[i] = (sum_k in_k[i]) / (0.0001f + n_connected), out is
 * expected pre-zeroed (node.rs:288). Returns `present`. */
int orc_collect_and_average(float *out, const float *const *ins, int n_ins, size_t buf_size);
/* The f32 divisor collect_and_average ends with for n connected pipes. */
float orc_link_divisor(uint64_t n_connected);
/* derive lib.rs:135-153: control-port helper.  `ctl` NULL => port absent =>
 * fill with *atomic.  Otherwise map [-1,1] -> [lo,hi] per sample and latch
 * out[0] into *atomic. */
void orc_slider_input(float *out, const float *ctl, float lo, float hi, float *atomic, size_t n);

/* SimpleNode::process for one 128-frame block of one channel.
 *   in_a : "in" (or "a" for Add/Mix), in_b : "b" (Add/Mix) else NULL
 *   ctl  : per-slider control-port blocks in field order (may be NULL / hold
 *          NULLs for unconnected ports)                                     */
void orc_node_process(orc_node *n, const float *in_a, const float *in_b,
                      const float *const *ctl, float *out, size_t n_frames);

/* A linear chain Input->n0->n1->...->Output for one channel, evaluated the way
 * the reference does it: block-of-128 at a time, node at a time, each hop
 * going through collect_and_average with one connected pipe when `link_scale`
 * (node.rs:290-299), outputs pre-zeroed (node.rs:272).  `side` feeds port "b"
 * of Add/Mix nodes (NULL => unconnected => zeros).  n_frames must be a multiple
 * of block (block is normally ORC_BUF_SIZE). */
void orc_chain_run(orc_node **nodes, int n_nodes, int link_scale, const float *in,
                   const float *side, float *out, size_t n_frames, size_t block);

/* Same with control ports: ctl[3*k + j] is the signal feeding the j-th `as_input` slider port of
 * node k (field order; NULL = unconnected), n_frames long.  A connected control port is a link like
 * any other: its value passes collect_and_average when link_flags has bit0. */
void orc_chain_run_ctl(orc_node **nodes, int n_nodes, int link_flags, const float *in, const float *side,
                       const float *const *ctl, float *out, size_t n_frames, size_t block);

/* ---- synthetic input --------------------------------------------------- */
/* x[c,n] = (float)(h>>8) * 2^-23 - 1, h = fmix32(seed ^ c*0x9E3779B9 ^ n*0x85EBCA6B) */
float orc_noise(uint32_t seed, uint32_t channel, uint32_t n_abs);

/* ---- multi-channel drivers (cpu_baseline + big parity checks) ---------- */
/* Runs `n_channels` independent copies of the chain described by `proto`
 * (cloned per channel, fresh state) over `n_blocks` blocks of hashed noise,
 * channels [c0, c0+n_channels).  If out != NULL it receives [n_frames][n_channels]
 * frame-major f32.  If mix != NULL it receives the f64-accumulated per-frame sum
 * over channels of the final output (caller applies the link divisor).
 * n_threads<=1 => scalar single thread. Returns 0. */
/* CPU timing leg of bench.py (cpu_baseline): the reference-structured chain over n_channels channels on n_threads
 * threads; node clones and the input noise table are set up before the timed region (dspfx_oracle.c). */
int orc_bench_chain(orc_node **proto, int n_nodes, int link_flags, uint32_t seed, uint32_t n_channels,
                    uint32_t n_blocks, uint32_t block, int n_threads, double *wall_seconds);
/* Effective CPU parallelism of this process: a register-only loop on n_threads threads (dspfx_oracle.c). */
int orc_bench_spin(int n_threads, uint64_t iters_per_thread, double *wall_seconds);
int orc_run_noise_channels(orc_node **proto, int n_nodes, int link_scale, uint32_t seed,
                           uint32_t c0, uint32_t n_channels, uint32_t n_abs0,
                           uint32_t n_blocks, uint32_t block, float *out, double *mix,
                           int n_threads);

orc_node *orc_node_clone(const orc_node *n);

#ifdef __cplusplus
}
#endif
#endif
