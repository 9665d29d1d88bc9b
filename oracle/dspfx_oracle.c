/*
 * dspfx_oracle.c -- CPU restatement of the reference hot path.  See the header
 * for status (TEST INFRASTRUCTURE; PARITY UNPINNED) and build flags.
 * Every function cites the reference lines it follows (paths relative to
 * /root/reference/).  f32 everywhere except Fir's f64 accumulation; no FMA.
 */
#define _POSIX_C_SOURCE 200809L /* pthread barriers, clock_gettime (the CPU timing leg at the end of this file) */
#include "dspfx_oracle.h"

#include <math.h>
#include <pthread.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>

/* ------------------------------------------------------------------ utils */

static uint32_t f32_bits(float x) {
    uint32_t u;
    memcpy(&u, &x, 4);
    return u;
}
static float bits_f32(uint32_t u) {
    float x;
    memcpy(&x, &u, 4);
    return x;
}

/* Rust f32::signum: 1.0 for +0.0 and positives, -1.0 for -0.0 and negatives, NaN for NaN. */
static float rs_signum(float x) {
    if (x != x) return x;
    return (f32_bits(x) >> 31) ? -1.0f : 1.0f;
}
/* Rust f32::clamp(min,max) */
static float rs_clamp(float x, float lo, float hi) {
    if (x < lo) x = lo;
    if (x > hi) x = hi;
    return x;
}
/* max_by(f32::total_cmp) over |x|: on non-negative floats (incl. +NaN) the
 * total order is the integer order of the bit patterns. */
static float max_abs_total(const float *v, size_t n) {
    uint32_t m = 0;
    for (size_t i = 0; i < n; i++) {
        uint32_t b = f32_bits(v[i]) & 0x7fffffffu;
        if (b >= m) m = b;
    }
    return bits_f32(m);
}
/* llvm.powi with constant exponent expands to repeated multiplication. */
static float powi2(float x) { return x * x; }
static float powi3(float x) { return (x * x) * x; }
static float powi4(float x) {
    float s = x * x;
    return s * s;
}

/* -------------------------------------------------------------- lifecycle */

/* biquad.rs:62-76 regenerate_filter: divide by a0, reset state, swap coeffs. */
static void biquad_regenerate(orc_node *n) {
    float a0 = n->p[0];
    n->bq_a1 = n->p[1] / a0;
    n->bq_a2 = n->p[2] / a0;
    n->bq_b0 = n->p[3] / a0;
    n->bq_b1 = n->p[4] / a0;
    n->bq_b2 = n->p[5] / a0;
    n->bq_x1 = n->bq_x2 = n->bq_y1 = n->bq_y2 = 0.0f; /* reset_state() */
}

orc_node *orc_node_new(int kind) {
    orc_node *n = (orc_node *)calloc(1, sizeof(orc_node));
    n->kind = kind;
    switch (kind) {
    case ORC_GAIN: /* gain.rs:21 default 1.0 */
        n->p[0] = 1.0f;
        break;
    case ORC_BIQUAD: /* biquad.rs:18-41 slider defaults; 48-60 initial_filter */
        n->p[0] = 1.0f;
        n->p[1] = -0.24f;
        n->p[2] = 0.0f;
        n->p[3] = 0.758f;
        n->p[4] = 0.0f;
        n->p[5] = 0.0f;
        n->bq_a1 = -0.24f;
        n->bq_a2 = 0.0f;
        n->bq_b0 = 0.758f;
        n->bq_b1 = 0.0f;
        n->bq_b2 = 0.0f;
        break;
    case ORC_LOW_PASS: /* low_pass.rs:20 */
    case ORC_HIGH_PASS: /* high_pass.rs:20 */
        n->p[0] = 0.5f;
        break;
    case ORC_REVERB: /* reverb.rs:29-38: seconds 0.5, decay 0.5; 44-52 make_buffer(): a ring of 128 -- NOT the half
                        second the slider shows: refresh_seconds has not run on a node fresh from the menu */
        n->p[0] = 0.5f;
        n->p[1] = 0.5f;
        n->seconds_given = 1;
        orc_reverb_set_len(n, 128);
        break;
    case ORC_DISTORT: /* distort.rs:46-50: level Default (0.0), mode SoftClip */
        n->p[0] = 0.0f;
        n->mode = ORC_DIST_SOFT_CLIP;
        break;
    case ORC_OVERDRIVE: /* overdrive.rs:21-28 all Default (0.0) */
        break;
    case ORC_CHEBYSHEV: /* chebyshev.rs:21-25 default 0.0 */
        break;
    case ORC_FIR: { /* fir.rs:55-65: taps [1.0], Balanced, empty deque */
        double one = 1.0;
        n->mode = ORC_FIR_BALANCED;
        orc_fir_set_taps(n, &one, 1);
        break;
    }
    case ORC_ADD:
        break;
    case ORC_MIX: /* mix.rs:22-28 default 0.5 */
        n->p[0] = 0.5f;
        break;
    case ORC_SIGNAL_GEN: /* signal_gen.rs:41-55: amplitude 0.5, frequency 100.0, Sine, clock Default (0.0) */
        n->p[0] = 0.5f;
        n->p[1] = 100.0f;
        n->mode = ORC_SIG_SINE;
        break;
    default:
        break;
    }
    return n;
}

void orc_node_free(orc_node *n) {
    if (!n) return;
    free(n->ring);
    free(n->taps);
    free(n->dq);
    free(n);
}

orc_node *orc_node_clone(const orc_node *src) {
    orc_node *n = (orc_node *)malloc(sizeof(orc_node));
    memcpy(n, src, sizeof(orc_node));
    if (src->ring) {
        n->ring = (float *)malloc(sizeof(float) * src->ring_len);
        memcpy(n->ring, src->ring, sizeof(float) * src->ring_len);
    }
    if (src->taps) {
        n->taps = (double *)malloc(sizeof(double) * src->n_taps);
        memcpy(n->taps, src->taps, sizeof(double) * src->n_taps);
    }
    if (src->dq) {
        n->dq = (double *)malloc(sizeof(double) * (src->dq_cap ? src->dq_cap : 1));
        memcpy(n->dq, src->dq, sizeof(double) * src->dq_cap);
    }
    return n;
}

/* `this.field = value` of the generated restore() (dsp-stuff-derive/src/lib.rs:300-312) / a field initialiser of new():
 * the plain store, NO hook. */
void orc_node_init_param(orc_node *n, int idx, float v) {
    if (idx < 0 || idx >= 8) return;
    n->p[idx] = v;
    if (n->kind == ORC_REVERB && idx == 1) n->seconds_given = v > 0.0f;   /* 0 in a descriptor: "not given", the explicit ring length stands */
}

/* The node's `after_settings_change` hook, where it has one (grep over nodes/: biquad.rs:15 and reverb.rs:19, no other):
 *   BiQuad::regenerate_filter (biquad.rs:62-76): coefficients / a0, reset_state, update_coefficients;
 *   Reverb::refresh_seconds   (reverb.rs:55-71): num_samples = max((seconds * 48000) as usize, 128) from the CURRENT
 *       seconds slider, a NEW ring of that length, zero-filled, swapped in under the mutex -- the old ring and every echo
 *       in it are gone.  The ring length is explicit in this restatement (rivulet's capacity rounding is not in the tree):
 *       with the seconds slider given (p[1] > 0) the new length is orc_delay_len(seconds, mode & 1), else the current one.
 * Called (a) by the generated render() when ANY widget of the node changed -- `changed |= true` per slider / select, then
 * `if changed { hook(self) }`, lib.rs:487-497, 521-525, 560-568, 570-578 -- i.e. by orc_node_set_param for EVERY slot of the node,
 * Reverb's decay included; (b) once by the generated restore() after all fields are set (lib.rs:319-337).  A node fresh
 * from the menu (NodeStatic::new) has NOT run it: a Reverb keeps make_buffer()'s 128-sample ring until its first change. */
void orc_node_after_settings_change(orc_node *n) {
    if (n->kind == ORC_BIQUAD) biquad_regenerate(n);
    if (n->kind == ORC_REVERB) orc_reverb_set_len(n, n->seconds_given ? orc_delay_len(n->p[1], n->mode & 1) : n->ring_len);
}

void orc_node_set_param(orc_node *n, int idx, float v) {
    if (idx < 0 || idx >= 8) return;
    n->p[idx] = v;                       /* lib.rs:487-492: the slider's Relaxed store */
    if (n->kind == ORC_REVERB && idx == 1) n->seconds_given = 1;   /* a stored 0.0 is a value: max(0, 128) = 128 samples (reverb.rs:58) */
    orc_node_after_settings_change(n);   /* lib.rs:560-568 */
}

void orc_node_set_mode(orc_node *n, int mode) { n->mode = mode; }

uint32_t orc_delay_len(float seconds, int page_round) {
    /* reverb.rs:58: ((seconds * 48000.0) as usize).max(128); `as usize` truncates, saturating */
    float s = seconds * 48000.0f;
    uint32_t d;
    if (!(s > 0.0f))
        d = 0;
    else if (s >= 4294967040.0f)
        d = 0xffffffffu;
    else
        d = (uint32_t)s;
    if (d < 128) d = 128;
    if (page_round) d = (d + 1023u) / 1024u * 1024u;
    return d;
}

/* make_buffer() (reverb.rs:44-52) is the SAME three rivulet calls as refresh_seconds (reverb.rs:60-68) with 128 for num_samples:
 * circular_buffer::<f32>(n), try_grant(n), release(view().len()).  Whatever a reading of rivulet makes of those calls it
 * makes of both: under the page-rounded reading (mode bit 0) a node fresh from the menu delays by orc_delay_len(0, 1) = 1024
 * samples, not 128 (one hypothesis, one answer).  Call after the mode is set. */
void orc_reverb_make_buffer(orc_node *n) {
    if (n->kind == ORC_REVERB) orc_reverb_set_len(n, orc_delay_len(0.0f, n->mode & 1));
}

void orc_reverb_set_len(orc_node *n, uint32_t d) {
    /* reverb.rs:60-68: new ring, grant, fill(0.0), release(view.len()) => D zeros queued */
    free(n->ring);
    n->ring = (float *)calloc(d ? d : 1, sizeof(float));
    n->ring_len = d;
    n->ring_pos = 0;
    n->ip[0] = d;
}

void orc_fir_set_taps(orc_node *n, const double *taps_reversed, uint32_t n_taps) {
    free(n->taps);
    n->taps = (double *)malloc(sizeof(double) * (n_taps ? n_taps : 1));
    memcpy(n->taps, taps_reversed, sizeof(double) * n_taps);
    n->n_taps = n_taps;
    n->ip[1] = n_taps;
    /* fir.rs:153-171: loading an impulse response replaces `taps` ONLY.  `state` (fir.rs:64-65) is never
     * cleared, so the deque -- and with it every sample the node has seen -- survives the reload; fir_process
     * below then pops at most one sample per step (fir.rs:195-197), i.e. a deque longer than the new tap count
     * STAYS longer and its OLDEST samples pair with the taps (a pure extra delay of len - T samples), and a
     * shorter one keeps growing front-aligned like the warm-up.  A fresh node starts with an empty deque. */
}

void orc_node_reset(orc_node *n) {
    n->bq_x1 = n->bq_x2 = n->bq_y1 = n->bq_y2 = 0.0f;
    n->z = 0.0f;
    n->clock = 0.0f;
    n->env = 0.0f; /* Detector::new: last_env_frame = EQUILIBRIUM */
    if (n->ring) memset(n->ring, 0, sizeof(float) * n->ring_len);
    n->ring_pos = 0;
    free(n->dq);
    n->dq = NULL;
    n->dq_cap = n->dq_head = n->dq_len = 0;
}

/* ---------------------------------------------------- wrapper semantics */

float orc_link_divisor(uint64_t n_connected) {
    /* node.rs:166,179: f32 num_frames = 0.0001; += 1.0 per connected pipe */
    float num_frames = 0.0001f;
    for (uint64_t i = 0; i < n_connected; i++) num_frames += 1.0f;
    return num_frames;
}

int orc_collect_and_average(float *out, const float *const *ins, int n_ins, size_t buf_size) {
    /* node.rs:162-194 */
    float num_frames = 0.0001f;
    int r = 0;
    for (int k = 0; k < n_ins; k++) {
        if (!ins[k]) continue; /* view shorter than buf_size => skipped (174-176) */
        r = 1;
        num_frames += 1.0f;
        for (size_t i = 0; i < buf_size; i++) out[i] += ins[k][i]; /* 181-183 */
    }
    for (size_t i = 0; i < buf_size; i++) out[i] /= num_frames; /* 189-191 */
    return r;
}

void orc_slider_input(float *out, const float *ctl, float lo, float hi, float *atomic, size_t n) {
    /* dsp-stuff-derive/src/lib.rs:135-153 */
    if (ctl) {
        for (size_t i = 0; i < n; i++) {
            float y = (ctl[i] + 1.0f) / 2.0f;
            float z = rs_clamp(y, 0.0f, 1.0f);
            out[i] = lo + (hi - lo) * z;
        }
        if (n) *atomic = out[0]; /* 148: latch first element */
    } else {
        float val = *atomic;
        for (size_t i = 0; i < n; i++) out[i] = val; /* 150-151 */
    }
}

/* ------------------------------------------------------------- effects */

/* distort.rs:53-61 */
static float clip(float s) {
    if (s < -1.0f)
        return -1.0f;
    else if (s > 1.0f)
        return 1.0f;
    else
        return s;
}
/* distort.rs:63-69 */
static float do_hard_clip(float sample, float level) {
    if (level < 0.001f) return sample;
    return clip(sample * level) / level;
}
/* distort.rs:71-86 */
static float do_soft_clip(float sample, float level) {
    if (level < 0.001f) return sample;
    float s = sample * level;
    if (s > 1.0f)
        s = 2.0f / 3.0f;
    else if (s >= -1.0f && s <= 1.0f)
        s = s - (powi3(s) / 3.0f);
    else
        s = -2.0f / 3.0f;
    return clip(s) / level;
}
/* distort.rs:96-102 */
static float do_recip_soft_clip(float sample, float level) {
    if (level < 0.001f) return sample;
    return rs_signum(sample) * (1.0f - 1.0f / (fabsf(sample) * level + 1.0f));
}
/* distort.rs:104-126 (glibc libm == Rust std on linux-gnu) */
static float do_tanh(float sample, float level) {
    if (level < 0.001f) return sample;
    return tanhf(sample * level);
}
static float do_sin(float sample, float level) {
    if (level < 0.001f) return sample;
    return sinf(sample * level);
}
static float do_atan(float sample, float level) {
    if (level < 0.001f) return sample;
    return atanf(sample * level);
}
/* distort.rs:128-134 */
static float do_sqr(float sample, float level) {
    if (level < 0.001f) return sample;
    return powi2(sample * level) * rs_signum(sample * level);
}
/* distort.rs:136-144 */
static float do_cheb_4(float sample, float level) {
    if (level < 0.001f) return sample;
    float v = sample * level;
    return 8.0f * powi4(v) - 8.0f * powi2(v) + 1.0f;
}
/* distort.rs:146-172; operates on exactly one reference block (BUF_SIZE) */
static void fuzz(const float *input, float *output, const float *level, size_t n) {
    float z[ORC_BUF_SIZE], y[ORC_BUF_SIZE];
    float mx = max_abs_total(input, n);
    for (size_t i = 0; i < n; i++) {
        float q = clip(input[i] * level[i]) / mx;
        /* (1.0 - q.copysign(-1.0).exp()).copysign(-1.0) */
        float e = expf(-fabsf(q));
        z[i] = -fabsf(1.0f - e);
    }
    float mz = max_abs_total(z, n);
    for (size_t i = 0; i < n; i++) y[i] = clip(z[i] * mx) / mz;
    float my = max_abs_total(y, n);
    for (size_t i = 0; i < n; i++) output[i] = y[i] * mx / my;
}
/* overdrive.rs:31-43 */
static float do_overdrive(float sample, float boost, float level, float drive) {
    if (level < 0.001f) return sample;
    const float FRAC_PI_4 = 0.785398163397448309615660845819875721f;
    const float FRAC_2_PI = 0.636619772367581343075535053490057448f;
    float a = sample * boost;
    float b = FRAC_PI_4 * a;
    float c = atanf(b);
    float d = FRAC_2_PI * c;
    float mix = drive * d + (1.0f - drive) * sample;
    return mix * level;
}
/* chebyshev.rs:28-42 */
static float do_chebyshev(float sample, float level_pos, float level_neg) {
    if (sample >= 0.0f) {
        if (level_pos < 0.001f) return sample;
        return tanhf(sample * level_pos) / tanhf(level_pos);
    } else {
        if (level_neg < 0.001f) return sample;
        return tanhf(sample * level_neg) / tanhf(level_neg);
    }
}

/* fir.rs:192-223 with an explicit model of std VecDeque<f64> so the a/b slice split (as_slices, 201-202) is
 * reproduced: capacity grows by doubling from 4 (RawVec amortised growth, 8-byte elements), push_back writes at
 * (head + len) % cap, pop_front advances head, and growing a WRAPPED deque moves its shorter part the way
 * VecDeque::handle_capacity_increase does (std >= 1.67; the reference pins a 2023+ nightly):
 *   A  head <= old_cap - len (contiguous): nothing moves, head stays;
 *   B  the wrapped-around front part [0, tail_len) is the shorter one and fits: it is copied behind old_cap;
 *   C  otherwise the part [head, old_cap) moves to the END of the new buffer and head follows it.
 * While warming up from empty no pop has happened, so head is 0 and A applies; B / C only arise after a tap
 * reload with a longer impulse response (orc_fir_set_taps keeps the deque). */
static void fir_push_back(orc_node *n, double v) {
    if (n->dq_len == n->dq_cap) {
        const uint32_t old_cap = n->dq_cap;
        const uint32_t ncap = old_cap ? old_cap * 2 : 4;
        double *nd = (double *)malloc(sizeof(double) * ncap);
        if (old_cap) memcpy(nd, n->dq, sizeof(double) * old_cap);   /* realloc keeps the old slots in place */
        if (n->dq_head <= old_cap - n->dq_len) {
            /* A */
        } else {
            const uint32_t head_len = old_cap - n->dq_head;
            const uint32_t tail_len = n->dq_len - head_len;
            if (head_len > tail_len && ncap - old_cap >= tail_len) {
                memcpy(nd + old_cap, nd, sizeof(double) * tail_len);                 /* B */
            } else {
                const uint32_t new_head = ncap - head_len;
                memmove(nd + new_head, nd + n->dq_head, sizeof(double) * head_len);  /* C */
                n->dq_head = new_head;
            }
        }
        free(n->dq);
        n->dq = nd;
        n->dq_cap = ncap;
    }
    n->dq[(n->dq_head + n->dq_len) % n->dq_cap] = v;
    n->dq_len++;
}
static void fir_process(orc_node *n, const float *in, float *out, size_t n_frames) {
    /* fir.rs:187-190 */
    float divisor = (n->mode == ORC_FIR_AVERAGE) ? 1.0f / (float)n->n_taps : 1.0f;
    const uint32_t T = n->n_taps;
    for (size_t i = 0; i < n_frames; i++) {
        fir_push_back(n, (double)in[i]); /* 193 */
        if (n->dq_len > T) {             /* 195-197 pop_front */
            n->dq_head = (n->dq_head + 1) % n->dq_cap;
            n->dq_len--;
        }
        /* 201-202 as_slices */
        uint32_t n_a = n->dq_len, n_b = 0;
        if (n->dq_head + n->dq_len > n->dq_cap) {
            n_a = n->dq_cap - n->dq_head;
            n_b = n->dq_len - n_a;
        }
        /* 204-206: zip(a, taps) stops at the shorter */
        double acc = 0.0;
        uint32_t la = n_a < T ? n_a : T;
        for (uint32_t k = 0; k < la; k++) acc += n->dq[n->dq_head + k] * n->taps[k];
        float a = (float)acc;
        float b = 0.0f;
        if (n_a < T) { /* 208-214 */
            double accb = 0.0;
            uint32_t lb = n_b < (T - n_a) ? n_b : (T - n_a);
            for (uint32_t k = 0; k < lb; k++) accb += n->dq[k] * n->taps[n_a + k];
            b = (float)accb;
        }
        float val = a + b;       /* 216 */
        out[i] = val * divisor;  /* 222 */
    }
}

void orc_node_process(orc_node *n, const float *in_a, const float *in_b,
                      const float *const *ctl, float *out, size_t nf) {
    static const float zeros[ORC_BUF_SIZE] = {0};
    const float *c0 = ctl ? ctl[0] : NULL;
    const float *c1 = ctl ? ctl[1] : NULL;
    const float *c2 = ctl ? ctl[2] : NULL;
    switch (n->kind) {
    case ORC_GAIN: { /* gain.rs:27-37 */
        float level[ORC_BUF_SIZE];
        orc_slider_input(level, c0, 0.0f, 10.0f, &n->p[0], nf);
        for (size_t i = 0; i < nf; i++) out[i] = in_a[i] * level[i];
        break;
    }
    case ORC_BIQUAD: { /* biquad.rs:81-88 -> biquad 0.4.2 DirectForm1::run (restated) */
        for (size_t i = 0; i < nf; i++) {
            float x = in_a[i];
            float y = n->bq_b0 * x + n->bq_b1 * n->bq_x1 + n->bq_b2 * n->bq_x2 -
                      n->bq_a1 * n->bq_y1 - n->bq_a2 * n->bq_y2;
            n->bq_x2 = n->bq_x1;
            n->bq_x1 = x;
            n->bq_y2 = n->bq_y1;
            n->bq_y1 = y;
            out[i] = y;
        }
        break;
    }
    case ORC_LOW_PASS: { /* low_pass.rs:28-41 */
        float ratio = n->p[0], z = n->z;
        for (size_t i = 0; i < nf; i++) {
            out[i] = in_a[i] * (1.0f - ratio) + ratio * z;
            z = out[i];
        }
        n->z = z;
        break;
    }
    case ORC_HIGH_PASS: { /* high_pass.rs:28-41 */
        float ratio = n->p[0], z = n->z;
        for (size_t i = 0; i < nf; i++) {
            z = in_a[i] * (1.0f - ratio) + ratio * z;
            out[i] = in_a[i] - z;
        }
        n->z = z;
        break;
    }
    case ORC_REVERB: { /* reverb.rs:76-110 */
        float decay = n->p[0];
        uint32_t D = n->ring_len, pos = n->ring_pos;
        if (nf <= D) { /* source.try_grant(len) succeeds: D samples always queued */
            for (size_t i = 0; i < nf; i++) out[i] = in_a[i] + n->ring[(pos + i) % D] * decay; /* 86-91 */
            /* release(len) then sink grant+copy+release (99-103): FIFO append */
            for (size_t i = 0; i < nf; i++) n->ring[(pos + i) % D] = out[i];
            n->ring_pos = (uint32_t)((pos + nf) % D);
        } else { /* 92-95: "Reverb buffer is empty" */
            memcpy(out, in_a, sizeof(float) * nf);
        }
        break;
    }
    case ORC_DISTORT: { /* distort.rs:176-194 */
        float level[ORC_BUF_SIZE];
        orc_slider_input(level, c0, 0.0f, 30.0f, &n->p[0], nf);
        switch (n->mode) {
        case ORC_DIST_HARD_CLIP:
            for (size_t i = 0; i < nf; i++) out[i] = do_hard_clip(in_a[i], level[i]);
            break;
        case ORC_DIST_SOFT_CLIP:
            for (size_t i = 0; i < nf; i++) out[i] = do_soft_clip(in_a[i], level[i]);
            break;
        case ORC_DIST_TANH:
            for (size_t i = 0; i < nf; i++) out[i] = do_tanh(in_a[i], level[i]);
            break;
        case ORC_DIST_RECIP_SOFT_CLIP:
            for (size_t i = 0; i < nf; i++) out[i] = do_recip_soft_clip(in_a[i], level[i]);
            break;
        case ORC_DIST_FUZZ:
            fuzz(in_a, out, level, nf);
            break;
        case ORC_DIST_SIN:
            for (size_t i = 0; i < nf; i++) out[i] = do_sin(in_a[i], level[i]);
            break;
        case ORC_DIST_ATAN:
            for (size_t i = 0; i < nf; i++) out[i] = do_atan(in_a[i], level[i]);
            break;
        case ORC_DIST_SQUARE:
            for (size_t i = 0; i < nf; i++) out[i] = do_sqr(in_a[i], level[i]);
            break;
        case ORC_DIST_CHEBYSHEV4:
            for (size_t i = 0; i < nf; i++) out[i] = do_cheb_4(in_a[i], level[i]);
            break;
        default:
            memcpy(out, in_a, sizeof(float) * nf);
        }
        break;
    }
    case ORC_OVERDRIVE: { /* overdrive.rs:58-72; slider field order boost, drive, level */
        float boost[ORC_BUF_SIZE], level[ORC_BUF_SIZE], drive[ORC_BUF_SIZE];
        orc_slider_input(boost, c0, 0.0f, 30.0f, &n->p[0], nf);
        orc_slider_input(level, c2, 0.0f, 1.0f, &n->p[2], nf);
        orc_slider_input(drive, c1, 0.0f, 1.0f, &n->p[1], nf);
        for (size_t i = 0; i < nf; i++) out[i] = do_overdrive(in_a[i], boost[i], level[i], drive[i]);
        break;
    }
    case ORC_CHEBYSHEV: { /* chebyshev.rs:52-62 */
        for (size_t i = 0; i < nf; i++) out[i] = do_chebyshev(in_a[i], n->p[0], n->p[1]);
        break;
    }
    case ORC_FIR:
        fir_process(n, in_a, out, nf);
        break;
    case ORC_ADD: { /* add.rs:26-33 */
        const float *b = in_b ? in_b : zeros;
        for (size_t i = 0; i < nf; i++) out[i] = in_a[i] + b[i];
        break;
    }
    case ORC_SIGNAL_GEN: { /* signal_gen.rs:57-129: a source; sliders amplitude -1..=1, frequency 0.1..=20000 */
        float amplitude[ORC_BUF_SIZE], frequency[ORC_BUF_SIZE];
        const float TAU = 6.28318530717958647692528676655900577f;
        orc_slider_input(amplitude, c0, -1.0f, 1.0f, &n->p[0], nf);
        orc_slider_input(frequency, c1, 0.1f, 20000.0f, &n->p[1], nf);
        if (n->mode == ORC_SIG_CONSTANT) { /* do_const, 106-108: clock untouched */
            for (size_t i = 0; i < nf; i++) out[i] = amplitude[i];
            break;
        }
        float clock = n->clock, total = 0.0f;
        for (size_t i = 0; i < nf; i++) {
            float step = frequency[i] / 48000.0f;
            total += step;
            if (n->mode == ORC_SIG_SINE) /* 64 */
                out[i] = sinf((clock + total) * TAU) * amplitude[i];
            else if (n->mode == ORC_SIG_TRIANGLE) /* 80 */
                out[i] = (2.0f * fmodf(clock + total, 1.0f) - 1.0f) * amplitude[i];
            else /* Square, 96: compares the block-local `total`, not the phase */
                out[i] = (total > 0.5f ? 1.0f : -1.0f) * amplitude[i];
        }
        n->clock = fmodf(clock + total, 1.0f); /* 66-67 */
        break;
    }
    case ORC_ENVELOPE: { /* envelope.rs:34-52; dasp_envelope 0.11.0 detect/mod.rs (Detector::next), dasp_peak full_wave */
        /* set_attack_frames / set_release_frames every block (envelope.rs:45-46): calc_gain(n) */
        float ga = n->p[0] == 0.0f ? 0.0f : powf(2.71828182845904523536028747135266250f, -1.0f / n->p[0]);
        float gr = n->p[1] == 0.0f ? 0.0f : powf(2.71828182845904523536028747135266250f, -1.0f / n->p[1]);
        float l = n->env;
        for (size_t i = 0; i < nf; i++) {
            float x = in_a[i];
            float d = x < 0.0f ? -x : x;           /* full_wave: if s < EQUILIBRIUM { -s } else { s } */
            float gain = l < d ? ga : gr;          /* attack while the envelope is below the rectified sample */
            float diff = l + (-d);                 /* l.add_amp(-d) */
            l = d + diff * gain;                   /* d.add_amp(diff.mul_amp(gain)) */
            out[i] = l;
        }
        n->env = l;
        break;
    }
    case ORC_MIX: { /* mix.rs:33-46 */
        float ratio[ORC_BUF_SIZE];
        const float *b = in_b ? in_b : zeros;
        orc_slider_input(ratio, c0, 0.0f, 1.0f, &n->p[0], nf);
        for (size_t i = 0; i < nf; i++) out[i] = (b[i] * ratio[i]) + (in_a[i] * (1.0f - ratio[i]));
        break;
    }
    default:
        memcpy(out, in_a, sizeof(float) * nf);
    }
}

/* ---------------------------------------------------------------- chain */

void orc_chain_run(orc_node **nodes, int n_nodes, int link_flags, const float *in,
                   const float *side, float *out, size_t n_frames, size_t block) {
    orc_chain_run_ctl(nodes, n_nodes, link_flags, in, side, NULL, out, n_frames, block);
}

void orc_chain_run_ctl(orc_node **nodes, int n_nodes, int link_flags, const float *in, const float *side,
                       const float *const *ctl, float *out, size_t n_frames, size_t block) {
    float cur[ORC_BUF_SIZE], port_a[ORC_BUF_SIZE], port_b[ORC_BUF_SIZE], obuf[ORC_BUF_SIZE];
    float port_c[3][ORC_BUF_SIZE];
    if (block == 0 || block > ORC_BUF_SIZE) block = ORC_BUF_SIZE;
    for (size_t f0 = 0; f0 < n_frames; f0 += block) {
        size_t nf = n_frames - f0 < block ? n_frames - f0 : block;
        memcpy(cur, in + f0, sizeof(float) * nf);
        for (int k = 0; k < n_nodes; k++) {
            /* bit0: hops between chain nodes; bit1: the hop into the first node too */
            int scale = (k == 0) ? (link_flags & 2) : (link_flags & 1);
            const float *pa = cur, *pb = NULL;
            if (scale) { /* node.rs:288,290-299: zeroed port buffer + collect_and_average */
                const float *ins[1] = {cur};
                memset(port_a, 0, sizeof(float) * nf);
                orc_collect_and_average(port_a, ins, 1, nf);
                pa = port_a;
            }
            if (side && (nodes[k]->kind == ORC_ADD || nodes[k]->kind == ORC_MIX)) {
                pb = side + f0;
                if (link_flags & 1) {
                    const float *ins[1] = {side + f0};
                    memset(port_b, 0, sizeof(float) * nf);
                    orc_collect_and_average(port_b, ins, 1, nf);
                    pb = port_b;
                }
            }
            const float *cp[3] = {NULL, NULL, NULL};
            if (ctl) {
                for (int j = 0; j < 3; j++) {
                    const float *sig = ctl[3 * k + j];
                    if (!sig) continue;
                    cp[j] = sig + f0;
                    if (link_flags & 1) { /* a control link is averaged like any other port (node.rs:290-299) */
                        const float *ins[1] = {sig + f0};
                        memset(port_c[j], 0, sizeof(float) * nf);
                        orc_collect_and_average(port_c[j], ins, 1, nf);
                        cp[j] = port_c[j];
                    }
                }
            }
            memset(obuf, 0, sizeof(float) * nf); /* node.rs:272 */
            orc_node_process(nodes[k], pa, pb, ctl ? cp : NULL, obuf, nf);
            memcpy(cur, obuf, sizeof(float) * nf); /* node.rs:321-325 fan-out copy */
        }
        memcpy(out + f0, cur, sizeof(float) * nf);
    }
}

/* ---------------------------------------------------------------- noise */

float orc_noise(uint32_t seed, uint32_t channel, uint32_t n_abs) {
    uint32_t h = seed ^ (channel * 0x9E3779B9u) ^ (n_abs * 0x85EBCA6Bu);
    h ^= h >> 16;
    h *= 0x85EBCA6Bu;
    h ^= h >> 13;
    h *= 0xC2B2AE35u;
    h ^= h >> 16;
    return (float)(h >> 8) * 0x1p-23f - 1.0f;
}

/* ------------------------------------------------ multi-channel drivers */

typedef struct {
    orc_node **proto;
    int n_nodes, link_flags;
    uint32_t seed, c0, c_begin, c_end, n_channels, n_abs0, n_blocks, block;
    float *out;
    double *mix; /* per-thread partial [n_blocks*block] or NULL */
} noise_job;

static void *noise_worker(void *arg) {
    noise_job *j = (noise_job *)arg;
    size_t nf = (size_t)j->n_blocks * j->block;
    float *x = (float *)malloc(sizeof(float) * j->block);
    float *y = (float *)malloc(sizeof(float) * j->block);
    orc_node **nodes = (orc_node **)malloc(sizeof(orc_node *) * (size_t)j->n_nodes);
    for (uint32_t ci = j->c_begin; ci < j->c_end; ci++) {
        for (int k = 0; k < j->n_nodes; k++) nodes[k] = orc_node_clone(j->proto[k]);
        for (uint32_t b = 0; b < j->n_blocks; b++) {
            uint32_t n0 = j->n_abs0 + b * j->block;
            for (uint32_t f = 0; f < j->block; f++) x[f] = orc_noise(j->seed, j->c0 + ci, n0 + f);
            orc_chain_run(nodes, j->n_nodes, j->link_flags, x, NULL, y, j->block, j->block);
            if (j->out)
                for (uint32_t f = 0; f < j->block; f++)
                    j->out[((size_t)b * j->block + f) * j->n_channels + ci] = y[f];
            if (j->mix)
                for (uint32_t f = 0; f < j->block; f++) j->mix[(size_t)b * j->block + f] += (double)y[f];
        }
        for (int k = 0; k < j->n_nodes; k++) orc_node_free(nodes[k]);
    }
    (void)nf;
    free(nodes);
    free(x);
    free(y);
    return NULL;
}

int orc_run_noise_channels(orc_node **proto, int n_nodes, int link_flags, uint32_t seed,
                           uint32_t c0, uint32_t n_channels, uint32_t n_abs0,
                           uint32_t n_blocks, uint32_t block, float *out, double *mix,
                           int n_threads) {
    if (n_threads < 1) n_threads = 1;
    if ((uint32_t)n_threads > n_channels) n_threads = (int)(n_channels ? n_channels : 1);
    if (block == 0 || block > ORC_BUF_SIZE) block = ORC_BUF_SIZE;
    size_t nf = (size_t)n_blocks * block;
    noise_job *jobs = (noise_job *)calloc((size_t)n_threads, sizeof(noise_job));
    pthread_t *th = (pthread_t *)calloc((size_t)n_threads, sizeof(pthread_t));
    for (int t = 0; t < n_threads; t++) {
        noise_job *j = &jobs[t];
        j->proto = proto;
        j->n_nodes = n_nodes;
        j->link_flags = link_flags;
        j->seed = seed;
        j->c0 = c0;
        j->n_channels = n_channels;
        j->c_begin = (uint32_t)((uint64_t)n_channels * (uint64_t)t / (uint64_t)n_threads);
        j->c_end = (uint32_t)((uint64_t)n_channels * (uint64_t)(t + 1) / (uint64_t)n_threads);
        j->n_abs0 = n_abs0;
        j->n_blocks = n_blocks;
        j->block = block;
        j->out = out;
        j->mix = mix ? (double *)calloc(nf, sizeof(double)) : NULL;
    }
    if (n_threads == 1) {
        noise_worker(&jobs[0]);
    } else {
        for (int t = 0; t < n_threads; t++) pthread_create(&th[t], NULL, noise_worker, &jobs[t]);
        for (int t = 0; t < n_threads; t++) pthread_join(th[t], NULL);
    }
    if (mix) {
        for (size_t i = 0; i < nf; i++) mix[i] = 0.0;
        for (int t = 0; t < n_threads; t++) {
            for (size_t i = 0; i < nf; i++) mix[i] += jobs[t].mix[i];
            free(jobs[t].mix);
        }
    }
    free(jobs);
    free(th);
    return 0;
}

/* ---------------------------------------------------------- CPU timing leg (bench.py: cpu_baseline, kind "port")
 * The reference-structured chain (orc_chain_run: gather, /1.0001, process, scatter per node, 128-frame blocks,
 * node.rs:267-352) over n_channels independent channels on n_threads threads, timed INSIDE this function between a
 * start barrier and the last thread's finish.  Nothing but the chain is in the timed loops: every thread clones the
 * nodes once before the barrier and resets them per channel (orc_node_reset: the zeroing a fresh node's state costs
 * anyway, without the allocator), and the input is a per-thread table of hashed noise blocks filled before the barrier
 * (the GPU path reads pre-materialised input too).  *wall_seconds = barrier release -> last thread done. */
enum { ORC_BENCH_TABLE_BLOCKS = 61 }; /* prime: channel ci starts at block (7 ci) mod 61 of its thread's table */
typedef struct {
    orc_node **proto;
    int n_nodes, link_flags;
    uint32_t seed, c_begin, c_end, n_blocks, block;
    pthread_barrier_t *bar;
    double sink;
} bench_job;

static double now_seconds(void) {
    struct timespec ts;
    clock_gettime(CLOCK_MONOTONIC, &ts);
    return (double)ts.tv_sec + 1e-9 * (double)ts.tv_nsec;
}

static void *bench_worker(void *arg) {
    bench_job *j = (bench_job *)arg;
    const uint32_t B = j->block;
    float *tab = (float *)malloc(sizeof(float) * B * ORC_BENCH_TABLE_BLOCKS);
    float y[ORC_BUF_SIZE];
    orc_node **nodes = (orc_node **)malloc(sizeof(orc_node *) * (size_t)j->n_nodes);
    for (int k = 0; k < j->n_nodes; k++) nodes[k] = orc_node_clone(j->proto[k]);
    for (uint32_t i = 0; i < B * ORC_BENCH_TABLE_BLOCKS; i++) tab[i] = orc_noise(j->seed, j->c_begin, i);
    double sink = 0.0;
    pthread_barrier_wait(j->bar);
    for (uint32_t ci = j->c_begin; ci < j->c_end; ci++) {
        for (int k = 0; k < j->n_nodes; k++) orc_node_reset(nodes[k]);
        uint32_t tb = (7u * ci) % ORC_BENCH_TABLE_BLOCKS;
        for (uint32_t b = 0; b < j->n_blocks; b++) {
            orc_chain_run(nodes, j->n_nodes, j->link_flags, tab + (size_t)tb * B, NULL, y, B, B);
            sink += (double)y[B - 1];
            tb = tb + 1 == ORC_BENCH_TABLE_BLOCKS ? 0 : tb + 1;
        }
    }
    j->sink = sink;
    for (int k = 0; k < j->n_nodes; k++) orc_node_free(nodes[k]);
    free(nodes);
    free(tab);
    return NULL;
}

int orc_bench_chain(orc_node **proto, int n_nodes, int link_flags, uint32_t seed, uint32_t n_channels,
                    uint32_t n_blocks, uint32_t block, int n_threads, double *wall_seconds) {
    if (n_threads < 1) n_threads = 1;
    if ((uint32_t)n_threads > n_channels) n_threads = (int)(n_channels ? n_channels : 1);
    if (block == 0 || block > ORC_BUF_SIZE) block = ORC_BUF_SIZE;
    bench_job *jobs = (bench_job *)calloc((size_t)n_threads, sizeof(bench_job));
    pthread_t *th = (pthread_t *)calloc((size_t)n_threads, sizeof(pthread_t));
    pthread_barrier_t bar;
    pthread_barrier_init(&bar, NULL, (unsigned)n_threads + 1u);
    for (int t = 0; t < n_threads; t++) {
        bench_job *j = &jobs[t];
        j->proto = proto;
        j->n_nodes = n_nodes;
        j->link_flags = link_flags;
        j->seed = seed;
        j->c_begin = (uint32_t)((uint64_t)n_channels * (uint64_t)t / (uint64_t)n_threads);
        j->c_end = (uint32_t)((uint64_t)n_channels * (uint64_t)(t + 1) / (uint64_t)n_threads);
        j->n_blocks = n_blocks;
        j->block = block;
        j->bar = &bar;
        pthread_create(&th[t], NULL, bench_worker, j);
    }
    pthread_barrier_wait(&bar); /* every thread has its nodes and its input table */
    const double t0 = now_seconds();
    double sink = 0.0;
    for (int t = 0; t < n_threads; t++) {
        pthread_join(th[t], NULL);
        sink += jobs[t].sink;
    }
    const double t1 = now_seconds();
    if (wall_seconds) *wall_seconds = t1 - t0;
    pthread_barrier_destroy(&bar);
    free(jobs);
    free(th);
    return sink == 12345.678 ? 1 : 0; /* keeps the outputs observable */
}

/* How many CPUs does this process really get?  n_threads threads each run the same register-only loop (the noise hash, no
 * memory traffic); aggregate rate / one thread's rate = the effective parallelism (a container's CPU quota or SMT sharing
 * shows here, memory bandwidth does not).  bench.py prints it next to the chain's scaling table. */
typedef struct {
    uint64_t iters;
    pthread_barrier_t *bar;
    uint32_t sink;
} spin_job;
static void *spin_worker(void *arg) {
    spin_job *j = (spin_job *)arg;
    uint32_t h = 12345u;
    pthread_barrier_wait(j->bar);
    for (uint64_t i = 0; i < j->iters; i++) {
        h ^= h >> 16;
        h *= 0x85EBCA6Bu;
        h ^= h >> 13;
        h *= 0xC2B2AE35u;
        h += (uint32_t)i;
    }
    j->sink = h;
    return NULL;
}
int orc_bench_spin(int n_threads, uint64_t iters_per_thread, double *wall_seconds) {
    if (n_threads < 1) n_threads = 1;
    spin_job *jobs = (spin_job *)calloc((size_t)n_threads, sizeof(spin_job));
    pthread_t *th = (pthread_t *)calloc((size_t)n_threads, sizeof(pthread_t));
    pthread_barrier_t bar;
    pthread_barrier_init(&bar, NULL, (unsigned)n_threads + 1u);
    for (int t = 0; t < n_threads; t++) {
        jobs[t].iters = iters_per_thread;
        jobs[t].bar = &bar;
        pthread_create(&th[t], NULL, spin_worker, &jobs[t]);
    }
    pthread_barrier_wait(&bar);
    const double t0 = now_seconds();
    uint32_t sink = 0;
    for (int t = 0; t < n_threads; t++) {
        pthread_join(th[t], NULL);
        sink ^= jobs[t].sink;
    }
    if (wall_seconds) *wall_seconds = now_seconds() - t0;
    pthread_barrier_destroy(&bar);
    free(jobs);
    free(th);
    return sink == 0x12345678u ? 1 : 0;
}

