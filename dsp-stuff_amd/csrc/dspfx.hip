// dspfx.hip -- host side of the C ABI declared in include/dspfx.h.
//
// Owns parameters, coefficients and every piece of DSP state in HBM, plans the
// chain into stages (fused chain kernel | Fuzz | FIR) and launches them on the
// caller's stream.  No CPU fallback exists: without a HIP device every entry
// point that needs one returns DSPFX_ERR_NO_DEVICE.
// This file: lifecycle, parameter stores (queued, thread-safe), run_subblock and the process calls.  Which kernels serve a
// chain is decided in plan.hip; the run-time compiler lives in jit.hip, placement tuning in placement.hip, the RCCL collective
// in comm.hip, the host-buffer pipeline in host_pipe.hip, state export / import and the utilities in state_util.hip;
// engine.h is what they share.
#include "engine.h"

using namespace dspfx;
using namespace dspfx_host;

namespace dspfx_host {

// The last error text is kept per engine AND per calling thread: a GUI thread whose slider store failed reads its own
// message, not the one the audio thread produced a microsecond later.
thread_local const dspfx_engine *tl_err_engine = nullptr;
thread_local std::string tl_err;

int fail(dspfx_engine *e, int code, const char *fmt, ...) {
    if (e) {
        char buf[512];
        va_list ap;
        va_start(ap, fmt);
        vsnprintf(buf, sizeof buf, fmt, ap);
        va_end(ap);
        tl_err = buf;
        tl_err_engine = e;
        std::lock_guard<std::mutex> lk(e->err_mu);
        e->err = buf;
    }
    return code;
}


void biquad_regenerate(Node &n) {   // biquad.rs:62-76
    const float a0 = n.d.params[0];
    n.a1 = n.d.params[1] / a0;
    n.a2 = n.d.params[2] / a0;
    n.b0 = n.d.params[3] / a0;
    n.b1 = n.d.params[4] / a0;
    n.b2 = n.d.params[5] / a0;
}

void free_ring(Node &n) {
    for (float *g : n.groups)
        if (g) (void)hipFree(g);
    n.groups.clear();
    if (n.d_groups) (void)hipFree(n.d_groups);
    n.d_groups = nullptr;
    n.table_cap = 0;
}

void free_node(Node &n) {
    free_ring(n);
    if (n.state) (void)hipFree(n.state);
    n.state = nullptr;
    n.state_bytes = 0;
    fir_free(n.fir);
    for (int k = 0; k < 3; ++k) {
        if (n.latch[k]) (void)hipFree(n.latch[k]);
        n.latch[k] = nullptr;
    }
    n.latch_valid = 0;
}

int alloc_node_state(dspfx_engine *e, Node &n) {
    const size_t N = e->desc.channels;
    size_t bytes = 0;
    switch (n.d.kind) {
    case DSPFX_BIQUAD: bytes = 4 * N * sizeof(float); break;
    case DSPFX_LOW_PASS:
    case DSPFX_HIGH_PASS:
    case DSPFX_SIGNAL_GEN:
    case DSPFX_ENVELOPE: bytes = N * sizeof(float); break;   // z / z / clock / env
    default: break;
    }
    if (n.state && n.state_bytes != bytes) {
        (void)hipFree(n.state);
        n.state = nullptr;
    }
    if (bytes && !n.state) HIPCHK(e, hipMalloc((void **)&n.state, bytes));
    n.state_bytes = bytes;
    if (bytes) HIPCHK(e, hipMemset(n.state, 0, bytes));
    n.pos = 0;
    n.zero_left = 0;
    if (n.d.kind == DSPFX_REVERB) {   // reverb.rs:55-71: a brand-new zero ring (setup time: the device is idle, see set_nodes)
        const uint32_t D = n.D;
        free_ring(n);
        n.D = 0;
        const int rc = ring_resize(e, (int)(&n - e->nodes.data()), D, nullptr);   // nothing is written: the ring starts as `zero_left = D`
        if (rc) return rc;
        const int rt = tune_ring(e, n);
        if (rt) return rt;
    }
    if (n.d.kind == DSPFX_FIR) {
        const int rc = fir_configure(n.fir, n.taps.data(), (uint32_t)n.taps.size(), n.d.mode, (uint32_t)N,
                                     e->desc.max_frames);
        if (rc != 0) return fail(e, rc, "FIR configure failed: %s", fir_last_error());
    }
    return DSPFX_OK;
}

int validate_node(dspfx_engine *e, const dspfx_node_desc &d) {
    if (d.kind < 0 || d.kind >= DSPFX_N_KINDS) return fail(e, DSPFX_ERR_INVALID, "unknown node kind %d", d.kind);
    if (d.kind == DSPFX_DISTORT && (d.mode < 0 || d.mode > DSPFX_DIST_CHEBYSHEV4))
        return fail(e, DSPFX_ERR_INVALID, "unknown distort mode %d", d.mode);
    if (d.kind == DSPFX_SIGNAL_GEN && (d.mode < 0 || d.mode > DSPFX_SIG_CONSTANT))
        return fail(e, DSPFX_ERR_INVALID, "unknown signal generator mode %d", d.mode);
    if (d.kind == DSPFX_REVERB && d.delay_len < DSPFX_BUF_SIZE)
        return fail(e, DSPFX_ERR_INVALID, "delay_len %u < 128 (reverb.rs:58 clamps to >= 128)", d.delay_len);
    if (d.kind == DSPFX_FIR && (d.n_taps == 0 || !d.taps))
        return fail(e, DSPFX_ERR_INVALID, "FIR node needs taps");
    return DSPFX_OK;
}

void fill_slot(const dspfx_engine *e, int idx, SlotArgs &s, uint32_t nframes) {
    const Node &n = e->nodes[idx];
    memset(&s, 0, sizeof s);
    s.kind = n.d.kind;
    s.mode = n.d.mode;
    s.state = n.state;
    s.groups = n.d_groups;
    s.D = n.D;
    s.pos = n.pos;
    s.zero_rows = n.zero_left;
    if (n.d.kind == DSPFX_REVERB && nframes <= RING_GROUP_ROWS && n.D) {
        // the block's rows lie in at most three groups: the one row `pos` is in, the next, and group 0 past the wrap -- named
        // here so that no wave has to read the group table before it can form its first tap address
        if (n.probe_group) {
            s.g_a = s.g_b = s.g_0 = n.probe_group;
            s.g_ia = 0;
            s.g_valid = 1;
        } else if (!n.groups.empty()) {
            const uint32_t gi = n.pos >> 7, glast = (n.D - 1) >> 7;
            s.g_a = n.groups[gi];
            s.g_b = n.groups[gi < glast ? gi + 1 : glast];
            s.g_0 = n.groups[0];
            s.g_ia = gi;
            s.g_valid = 1;
        }
    }
    s.hop = node_hop(e, idx);
    s.rc = n.d.kind == DSPFX_DISTORT ? 1.0 / (double)n.d.params[0] : 0.0;
    for (int k = 0; k < 3; ++k) {
        s.ctl[k] = n.ctl_now[k];
        s.latch[k] = n.latch[k];
    }
    s.latch_valid = n.latch_valid;
    switch (n.d.kind) {
    case DSPFX_BIQUAD:
        s.p[0] = n.a1; s.p[1] = n.a2; s.p[2] = n.b0; s.p[3] = n.b1; s.p[4] = n.b2;
        break;
    case DSPFX_ENVELOPE:   // dasp_envelope calc_gain: the host's powf is the libm the reference calls
        for (int k = 0; k < 2; ++k)
            s.p[k] = n.d.params[k] == 0.0f ? 0.0f : powf(2.71828182845904523536028747135266250f, -1.0f / n.d.params[k]);
        break;
    default:
        for (int k = 0; k < 6; ++k) s.p[k] = n.d.params[k];
    }
}

// Delay-ring rows [r0, r0+nrows) (mod D) <-> dense host rows [nrows][N], through a device bounce buffer and a
// gather/scatter kernel, so the exported form is canonical whatever the ring's internal layout.
int ring_rows_copy(dspfx_engine *e, Node &n, uint32_t r0, uint32_t nrows, char *host, bool to_host) {
    const size_t N = e->desc.channels;
    const uint32_t W = e->desc.tile_channels ? e->desc.tile_channels : (uint32_t)N;
    const uint32_t chunk = (uint32_t)std::max<size_t>(1, std::min<size_t>(nrows, ((size_t)64 << 20) / (N * sizeof(float))));
    float *bounce = nullptr;
    HIPCHK(e, hipMalloc((void **)&bounce, (size_t)chunk * N * sizeof(float)));
    int rc = DSPFX_OK;
    for (uint32_t k = 0; k < nrows && rc == DSPFX_OK; k += chunk) {
        const uint32_t nr = std::min(chunk, nrows - k);
        const size_t bytes = (size_t)nr * N * sizeof(float);
        const uint32_t r = (uint32_t)(((uint64_t)r0 + k) % n.D);
        hipError_t err = hipSuccess;
        if (!to_host) err = hipMemcpy(bounce, host + (size_t)k * N * sizeof(float), bytes, hipMemcpyHostToDevice);
        if (err == hipSuccess) {
            launch_ring_copy(n.d_groups, bounce, (unsigned)N, W, n.D, r, nr, to_host, nullptr);
            err = hipGetLastError();
        }
        if (err == hipSuccess) err = hipDeviceSynchronize();
        if (err == hipSuccess && to_host) err = hipMemcpy(host + (size_t)k * N * sizeof(float), bounce, bytes, hipMemcpyDeviceToHost);
        if (err != hipSuccess) rc = fail(e, DSPFX_ERR_HIP, "ring copy: %s", hipGetErrorString(err));
    }
    (void)hipFree(bounce);
    return rc;
}

// rows of LDS the interpreter keeps per node (kind_nstate in chain_kernels.hip.h)
int state_rows(const Node &n) { return kind_nstate(n.d.kind); }

// One sub-block (nframes <= every delay length) through all stages.
int run_subblock(dspfx_engine *e, const float *in, const float *side, float *out, float *mix,
                 uint32_t nframes, uint32_t tile_frames, hipStream_t stream) {
    const uint32_t N = e->desc.channels;
    const float *src = in;
    Layout lay{};
    if (e->desc.tile_channels) {
        const uint32_t W = e->desc.tile_channels;
        lay.w_shift = (unsigned)__builtin_ctz(W);
        lay.w_mask = W - 1;
        lay.ld = W;
        lay.tile_stride = (size_t)tile_frames * W;
    } else {
        lay.w_shift = 31;
        lay.w_mask = 0x7fffffffu;
        lay.ld = N;
        lay.tile_stride = 0;
    }
    bool bus_done = false;   // a FIR node that ends the chain left the mix bus' partial sums itself
    for (size_t si = 0; si < e->stages.size(); ++si) {
        const Stage &st = e->stages[si];
        const bool last = si + 1 == e->stages.size();
        if (st.type == ST_FUSED) {
            if (st.count == 0 && (bus_done || !(last && (mix || e->partials_override || e->mp_building))) && src == out) continue;   // nothing to do
            GraphArgs ga;                   // a graph kernel reads the slots beyond ChainArgs from it
            memset(&ga, 0, sizeof ga);
            if (e->graph_mode) {
                const int n_nodes = (int)e->nodes.size();
                for (const dspfx_graph_link &l : e->wiring) {
                    const int blk = graph_input_block(l.src);
                    if (blk == 1 && !side) return fail(e, DSPFX_ERR_INVALID, "this graph reads a second block (DSPFX_GRAPH_INPUT2): side must not be null");
                    if (blk >= 2 && !e->io_in[blk]) return fail(e, DSPFX_ERR_INVALID, "this graph reads input block %d: pass it with dspfx_process_io", blk);
                    if (l.dst > n_nodes && !e->io_out[l.dst - n_nodes]) return fail(e, DSPFX_ERR_INVALID, "this graph writes output block %d: pass it with dspfx_process_io", l.dst - n_nodes);   // (blocks are contiguous: validate_graph)
                }
                for (int k = 2; k < GRAPH_IO; ++k) ga.xin[k - 2] = e->io_in[k] ? e->io_in[k] + e->io_off : nullptr;
                for (int m = 1; m < GRAPH_IO; ++m) ga.xout[m - 1] = e->io_out[m] ? e->io_out[m] + e->io_off : nullptr;
            }
            ChainArgs &a = ga.c;
            a.in = src;
            a.side = side;
            a.out = out;
            a.N = N;
            a.nframes = nframes;
            a.w_shift = lay.w_shift;
            a.w_mask = lay.w_mask;
            a.ld = lay.ld;
            a.io_tile_stride = lay.tile_stride;
            a.hop_div = e->hop_div;
            a.hop_rc = 1.0 / (double)e->hop_div;
            a.third_rc = 1.0 / 3.0;
            a.fast_div = st.fast_div ? 1 : 0;
            // DSPFX_XCD_REMAP=0 / 1 switches the XCD-contiguous block mapping off / on (A/B runs; EnvSwitches)
            const bool xcd_env = e->env.xcd_remap >= 0;
            a.xcd_remap = xcd_env ? e->env.xcd_remap : 1;
            a.n_slots = st.count;
            a.skip_store = (st.count == 0 && src == out) ? 1 : 0;   // an empty stage in place exists only for the mix bus
            // hop flag of the side input and of control links (both are ordinary links between nodes);
            // an unconnected side port reads zeros, for which the hop is a no-op
            a.side_hop = 0;
            if (e->desc.link_flags & DSPFX_LINK_INTERNAL) a.side_hop = (e->desc.link_flags & DSPFX_LINK_SIDE_RAW) ? 2 : 3;
            int rows = 0;
            for (int k = 0; k < st.count; ++k) {
                const Node &nk = e->nodes[st.first + k];
                if (nk.d.kind == DSPFX_REVERB && (!nk.d_groups || nk.groups.size() < ring_groups_for(nk.D) || !nk.D))
                    return fail(e, DSPFX_ERR_STATE, "REVERB node %d has no delay ring (an allocation failed earlier)", st.first + k);
                fill_slot(e, st.first + k, k < MAX_SLOTS ? a.slot[k] : ga.more[k - MAX_SLOTS], nframes);
                rows += state_rows(nk);
            }
            if (st.async || st.async_mod) adopt_async_jit(e, st);      // kernels the background compiler has finished: from this block on
            const Variant *v = st.var, *tail = e->tail;
            for (int k = 0; k < st.count; ++k) {   // modulated or latched sliders: only the MOD interpreter evaluates them
                const Node &nd = e->nodes[st.first + k];
                if (nd.latch_valid || nd.ctl_now[0] || nd.ctl_now[1] || nd.ctl_now[2]) {
                    // a run-time specialised kernel with control ports when one can be had (compiled on first use),
                    // else the control-port interpreter: two channels per lane above 131072 channels
                    // else the control-port interpreter: two channels per lane above 131072 channels -- or, while the
                    // specialised kernel is on its way, the channels per lane of THAT kernel (the same rows of bus partials)
                    if (!st.var_mod_tried) request_mod_kernel(e, st);
                    // (fixed when the job was submitted and kept when it comes back empty-handed: the bus' rows must not change in mid-stream)
                    const bool two = st.mod_two >= 0 ? st.mod_two == 1 : N > 131072u;
                    v = st.var_mod ? st.var_mod : ((e->dyn_mod2 && two && N % 2u == 0) ? e->dyn_mod2 : e->dyn_mod);
                    tail = e->tail_mod;
                }
            }
            // few channels, a whole 128-frame block, no control ports in play: four time slices per channel group
            if (st.var_ts && v == st.var && nframes == 4u * (uint32_t)st.var_ts->ts) {
                v = st.var_ts;
                // Its workgroups all issue their whole slices together; neighbouring channel groups on DIFFERENT XCDs (the
                // dispatcher's round robin) serve the HBM better than an eighth of the channels per XCD: config 2 25.6 -> 23.3 us
                // tiled, 24.2 -> 23.7 frame-major (profiles/r03_small_n.txt).  The large kernels keep the contiguous mapping.
                if (!xcd_env) a.xcd_remap = 0;
            }
            const uint32_t per_wave = 64u * v->cpl;
            // A few-channel engine (the time-sliced kernel is its main launch) whose N is not whole waves: two launches in a
            // row would both be latency-bound (11 + 11 us), so the guarded time-sliced kernel takes ALL channels in one
            // (N = 4099, 3-node chain: 61 us with the interpreter's one-wave launch behind the main one, 28 with the guarded
            // time-sliced launch behind it, ~12 alone).  Engines in channel windows keep the two launches.
            // Up to 32768 channels: one resident round of one-channel-per-lane workgroups, the same rows of bus partials.
            const bool whole_guard = v->ts && v->cpl == 1 && N < 32768u && st.var_ts_tail && tail == e->tail && N % per_wave && !e->win_n &&
                                     nframes == 4u * (uint32_t)st.var_ts_tail->ts;
            const uint32_t n_main = whole_guard ? 0u : N - N % per_wave;
            const uint32_t waves_main = n_main / per_wave;
            const bool deferred = last && e->partials_override != nullptr;
            a.mixpart = deferred ? e->partials_override : ((last && mix) ? e->mixpart : nullptr);
            const unsigned grid_main = v->ts ? waves_main : (waves_main * 64 + WG - 1) / WG;   // time-sliced: one workgroup per channel group
            // the bus' first stage leaves one row of partial sums per WORKGROUP (chain_kernels.hip.h, mixbus_flush); the
            // guarded tail launch runs one-wave workgroups.  Engines up to TS_MAX_CHANNELS -- the ones a time-sliced kernel
            // may serve -- leave one row per WAVE instead: one row per 64 x cpl channels in the interpreter, the specialised
            // standard kernel and the time-sliced kernels alike.  An engine starts on the interpreter and adopts the kernels
            // the background thread compiled at a block boundary nobody can predict (plan.hip): with the same rows before and
            // after -- and, above this size, the interpreter instantiation of the coming kernel's channels per lane -- the
            // bus' f32 summation order does not change at that switch (it used to).
            const bool rows_per_wave = !v->ts && N <= TS_MAX_CHANNELS;
            a.mix_per_wave = rows_per_wave ? 1 : 0;
            const unsigned rows_main = (v->ts || rows_per_wave) ? waves_main : grid_main;
            a.mix_stride = rows_main + (N - n_main + 63) / 64;
            if (a.mixpart && a.mix_stride > e->mixpart_cols) return fail(e, DSPFX_ERR_STATE, "mix partial buffer too small");
            // Same-block bus: the slice and final stages ride in the tail of this very launch (mix_tail) instead of two more
            // kernels behind it.  DSPFX_MIX_TAIL=0: the stand-alone kernels (A/B runs, tests: bit-identical).
            // (even block lengths: the tail reads two frames per lane with one 8-byte load)
            const bool tail_bus = last && mix && !deferred && !e->mp_building && e->mt_tickets && nframes % 2u == 0 && e->env.mix_tail != 0;
            if (tail_bus) {
                a.mt_tickets = e->mt_tickets;
                a.mt_part2 = e->mixpart_b;
                a.mt_mix = mix;
                a.mt_div = e->bus_div_now;
            }
            if (e->mp_building && last) {   // pipelined mix bus: earlier blocks' reductions ride in this launch
                const int cur = (int)(e->mp_count & 1), prev = cur ^ 1;
                a.mixpart = e->mixpart2[cur];
                if (a.mix_stride > e->mixpart_cols) return fail(e, DSPFX_ERR_STATE, "mix partial buffer too small");
                e->mp_rows[cur] = a.mix_stride;
                const int stage = (e->mp_count >= 1 ? 1 : 0) | (e->mp_count >= 2 ? 2 : 0);
                float *b_cur = cur ? e->mixpart_b2 : e->mixpart_b, *b_prev = cur ? e->mixpart_b : e->mixpart_b2;
                if (grid_main > MIX_SLICES) {
                    a.mp_stage = stage;
                    a.mp_rows_a = e->mp_rows[prev];
                    a.mp_prev_a = e->mixpart2[prev];
                    a.mp_cur_b = b_cur;
                    a.mp_prev_b = b_prev;
                    a.mp_mix = e->mp_mix_now;
                    a.mp_div = e->mp_div_now;
                } else {                    // too few workgroups to host the prologue: same stages as stand-alone kernels
                    if (stage & 2) {
                        launch_mix_reduce_final(b_prev, e->mp_mix_now, nframes, stream);
                        if (e->mp_div_now != 0.0f) launch_mix_finish(e->mp_mix_now, nframes, e->mp_div_now, stream);
                    }
                    if (stage & 1) launch_mix_reduce_slices(e->mixpart2[prev], b_cur, nframes, e->mp_rows[prev], stream);
                }
            }
            if (e->win_n) {   // a channel window (multiples of 1024 channels; the last one runs to N)
                const uint32_t w0 = e->win_c0, w1 = std::min(n_main, e->win_c0 + e->win_n);
                if (w1 > w0) {
                    a.c_base = w0;
                    a.n_launch = w1 - w0;
                    a.wave_base = (v->ts || rows_per_wave) ? w0 / per_wave : w0 / (per_wave * (WG / 64));   // rows of the windows before this one
                    ProfScope ps(e, si, stream);
                    if (launch_variant(v, a, v->ts ? (w1 - w0) / per_wave : ((w1 - w0) / v->cpl + WG - 1) / WG, WG, (unsigned)(rows * WG * v->cpl * sizeof(float)), stream)) {
                        if (tail_bus) (void)hipMemsetAsync(e->mt_tickets, 0, (MIX_SLICES + 1) * sizeof(unsigned), stream);
                        return fail(e, DSPFX_ERR_HIP, "kernel launch failed");
                    }
                }
            } else if (n_main) {
                a.c_base = 0;
                a.n_launch = n_main;
                a.wave_base = 0;
                const unsigned grid = grid_main;
                ProfScope ps(e, si, stream);
                if (launch_variant(v, a, grid, WG, (unsigned)(rows * WG * v->cpl * sizeof(float)), stream))
                    return fail(e, DSPFX_ERR_HIP, "kernel launch failed");
            }
            a.mp_stage = 0;   // the guarded tail launch never hosts the prologue
            if ((N % per_wave || whole_guard) && (!e->win_n || e->win_c0 + e->win_n >= N)) {   // ragged tail: guarded one-wave blocks, lane per channel
                const uint32_t n_tail = N - n_main;
                a.c_base = n_main;
                a.n_launch = n_tail;
                a.wave_base = rows_main;
                if (st.var_ts_tail && tail == e->tail && nframes == 4u * (uint32_t)st.var_ts_tail->ts) {
                    a.xcd_remap = 0;            // whole 128-frame blocks: four slices per 64 channels (pick_ts_tail_variant)
                    if (whole_guard) {          // the stage's only launch: it is the one dspfx_profile_read reports
                        ProfScope ps(e, si, stream);
                        (void)launch_variant(st.var_ts_tail, a, (n_tail + 63) / 64, WG, 0, stream);
                    } else {
                        (void)launch_variant(st.var_ts_tail, a, (n_tail + 63) / 64, WG, 0, stream);
                    }
                } else {
                    (void)launch_variant(tail, a, (n_tail + 63) / 64, 64, (unsigned)(rows * WG * sizeof(float)), stream);
                }
            }
            {
                const hipError_t lerr = hipGetLastError();
                if (lerr != hipSuccess) {
                    // one of the block's launches did not happen: the rows it would have ticketed never arrive, and the
                    // counters the other launch bumped would stay non-zero for ever -- every later bus silently stale
                    if (tail_bus) (void)hipMemsetAsync(e->mt_tickets, 0, (MIX_SLICES + 1) * sizeof(unsigned), stream);
                    return fail(e, DSPFX_ERR_HIP, "kernel launch failed: %s", hipGetErrorString(lerr));
                }
            }
            if (deferred) {
                e->part_stride[e->flip] = a.mix_stride;
                e->part_frames[e->flip] = nframes;
            } else if (a.mixpart && !e->mp_building && !tail_bus) {
                launch_mix_reduce(e->mixpart, e->mixpart_b, mix, nframes, a.mix_stride, stream);
                if (e->bus_div_now != 0.0f) launch_mix_finish(mix, nframes, e->bus_div_now, stream);
                HIPCHK(e, hipGetLastError());
            }
        } else if (st.type == ST_FUZZ) {
            const Node &n = e->nodes[st.first];
            FuzzArgs f{src, out, N, nframes, n.d.params[0], e->hop_div, node_hop(e, st.first),
                       (e->desc.link_flags & DSPFX_LINK_INTERNAL) ? 1 : 0, n.ctl_now[0], n.latch[0], n.latch_valid & 1, 0, lay};
            ProfScope ps(e, si, stream);
            launch_fuzz(f, stream);
            HIPCHK(e, hipGetLastError());
        } else {   // ST_FIR
            Node &n = e->nodes[st.first];
            hipEvent_t ea = nullptr, eb = nullptr;
            if (e->profiling) {
                ea = take_event(e);
                eb = take_event(e);
                if (e->prof.size() <= si) e->prof.resize(si + 1);
                e->prof[si].emplace_back(ea, eb);
            }
            // The FIR node ends the chain and the Output node's mix bus is wanted: the sweep's epilogue leaves the bus'
            // first-stage partials (one row per 32-channel tile) and the empty chain kernel that would otherwise read the
            // whole FIR output again just to sum it is skipped.  Slice / final stages as after a chain kernel; in the
            // pipelined form the sweep's first workgroups host them like a chain kernel's do.
            const bool ends_chain = si + 2 == e->stages.size() && e->stages[si + 1].type == ST_FUSED && e->stages[si + 1].count == 0 && !e->win_n;
            const bool want_bus = ends_chain && (mix || e->partials_override || e->mp_building);
            const uint32_t frows = (N + 31) / 32;
            const bool deferred = want_bus && e->partials_override != nullptr;
            float *fpart = nullptr;
            FirMixPipe fmp;      // the pipelined bus' stages of earlier blocks: hosted by the sweep or launched by fir_process
            if (want_bus) {
                if (frows > e->mixpart_cols) return fail(e, DSPFX_ERR_STATE, "mix partial buffer too small");
                fpart = deferred ? e->partials_override : e->mixpart;
                if (e->mp_building) {
                    const int cur = (int)(e->mp_count & 1), prev = cur ^ 1;
                    fpart = e->mixpart2[cur];
                    e->mp_rows[cur] = frows;
                    const int stage = (e->mp_count >= 1 ? 1 : 0) | (e->mp_count >= 2 ? 2 : 0);
                    float *b_cur = cur ? e->mixpart_b2 : e->mixpart_b, *b_prev = cur ? e->mixpart_b : e->mixpart_b2;
                    fmp.stage = stage;
                    fmp.rows_a = e->mp_rows[prev];
                    fmp.prev_a = e->mixpart2[prev];
                    fmp.cur_b = b_cur;
                    fmp.prev_b = b_prev;
                    fmp.mix = e->mp_mix_now;
                    fmp.div = e->mp_div_now;
                }
            }
            const int rc = fir_process(n.fir, src, out, nframes, node_hop(e, st.first), e->hop_div, lay, stream, ea, eb, fpart, &fmp);
            if (rc != 0) return fail(e, rc, "FIR: %s", fir_last_error());
            if (want_bus) {
                if (deferred) {
                    e->part_stride[e->flip] = frows;
                    e->part_frames[e->flip] = nframes;
                } else if (!e->mp_building) {
                    launch_mix_reduce(e->mixpart, e->mixpart_b, mix, nframes, frows, stream);
                    if (e->bus_div_now != 0.0f) launch_mix_finish(mix, nframes, e->bus_div_now, stream);
                    HIPCHK(e, hipGetLastError());
                }
                bus_done = true;
            }
        }
        src = out;
    }
    if (!e->win_last) return DSPFX_OK;   // more channel windows of this block follow
    for (Node &n : e->nodes)   // a connected control port leaves per-channel latched values behind
        for (int k = 0; k < 3; ++k)
            if (n.ctl_now[k]) n.latch_valid |= 1 << k;
    // advance the delay rings (FIFO: the block's rows now hold the newest samples)
    for (Node &n : e->nodes)
        if (n.d.kind == DSPFX_REVERB) {
            n.pos = (uint32_t)(((uint64_t)n.pos + nframes) % n.D);
            n.zero_left -= std::min(n.zero_left, nframes);   // rows written since the ring's last clear are real samples
        }
    return DSPFX_OK;
}

// ---- threads and streams ------------------------------------------------------------------------------------------
// The engine's DSP state is read at the start of a block's kernels and written at their end, so everything that writes
// that state must be ordered with the blocks in flight.  Two rules do it:
//   * every state write is queued on the stream the state was last used on (cur_stream), never on the null stream;
//   * a call that uses the state on ANOTHER stream first makes that stream wait for an event recorded on cur_stream.
// Callers' streams are usually non-blocking (torch's, the bench's): the null stream orders nothing against them.
int bind_stream(dspfx_engine *e, hipStream_t s) {
    if (e->cur_stream_set && e->cur_stream != s) {
        bool ok = true;
        if (!e->ev_order) ok = hipEventCreateWithFlags(&e->ev_order, hipEventDisableTiming) == hipSuccess;
        ok = ok && hipEventRecord(e->ev_order, e->cur_stream) == hipSuccess && hipStreamWaitEvent(s, e->ev_order, 0) == hipSuccess;
        if (!ok) {                      // the old stream may be gone (its owner destroyed it): everything it held has run or never will
            (void)hipGetLastError();
            HIPCHK(e, hipDeviceSynchronize());
        }
    }
    e->cur_stream = s;
    e->cur_stream_set = true;
    return DSPFX_OK;
}

// Setup-time calls that free or re-allocate state (chain / graph set, delay length, tap reload, state import / export):
// nothing of this engine may be in flight, on any stream.
int quiesce(dspfx_engine *e) {
    HIPCHK(e, hipSetDevice(e->device));
    HIPCHK(e, hipDeviceSynchronize());
    return DSPFX_OK;
}
// ... and what they queued on the null stream has finished before a non-blocking stream can touch the new state.
int settle_null_stream(dspfx_engine *e) {
    HIPCHK(e, hipStreamSynchronize(nullptr));
    return DSPFX_OK;
}

void publish_kinds(dspfx_engine *e) {
    std::vector<float *> drop;
    {
        std::lock_guard<std::mutex> lk(e->pend_mu);
        e->pub_kinds.clear();
        e->pub_rev.assign(e->nodes.size(), dspfx_engine::PubReverb{});
        for (size_t i = 0; i < e->nodes.size(); ++i) {
            const Node &n = e->nodes[i];
            e->pub_kinds.push_back(n.d.kind);
            if (n.d.kind == DSPFX_REVERB) e->pub_rev[i] = dspfx_engine::PubReverb{n.d.params[1], n.d.mode, n.D, n.groups.size(), n.seconds_given};
        }
        e->pending.clear();                 // stores aimed at the nodes that no longer exist
        ++e->chain_gen;                     // ... and ring capacity being allocated for them is dropped by its allocator
    }
    {
        std::lock_guard<std::mutex> lk(e->pool_mu);
        for (auto &pool : e->ring_pool)
            for (float *g : pool) drop.push_back(g);
        e->ring_pool.assign(e->nodes.size(), {});
    }
    for (float *g : drop) (void)hipFree(g);     // (never part of a ring: no block has seen them)
}

// One slider / mode store takes effect (api_mu held).  State writes go to `s` in stream order.  *replan: the stage split or a
// stage's division verdict changed.
int apply_store(dspfx_engine *e, const dspfx_engine::Store &st, std::vector<char> &biquad_reset, std::vector<char> &reverb_refresh, bool &replan) {
    if (st.node < 0 || st.node >= (int)e->nodes.size()) return DSPFX_OK;   // the chain was replaced since
    Node &n = e->nodes[(size_t)st.node];
    if (st.param < 0) {                                  // dspfx_set_mode
        n.d.mode = (int)st.value;
        if (n.d.kind == DSPFX_FIR) n.fir.mode = n.d.mode;
        replan = true;                                   // Fuzz <-> other modes changes the stage split
        return DSPFX_OK;
    }
    n.d.params[st.param] = st.value;
    if (n.d.kind == DSPFX_REVERB && st.param == 1) n.seconds_given = true;      // (0.0 included: a 128-sample ring, reverb.rs:58)
    if (st.param < 3) n.latch_valid &= ~(1 << st.param);   // a slider store overwrites the latched values
    if (n.d.kind == DSPFX_BIQUAD) {   // after_settings_change: renormalise + reset_state (biquad.rs:62-76)
        biquad_regenerate(n);
        biquad_reset[(size_t)st.node] = 1;
    }
    // Reverb's hook is attached to the NODE, not to the `seconds` slider: the generated render() runs it when ANY widget of
    // the node changed (reverb.rs:19, dsp-stuff-derive/src/lib.rs:487-497, 560-568), so a `decay` store swaps in a new
    // zero-filled ring too (reverb.rs:55-71).
    if (n.d.kind == DSPFX_REVERB) reverb_refresh[(size_t)st.node] = 1;
    if (n.d.kind == DSPFX_DISTORT) {
        // A new clip level is a new constant divisor.  Whether its fast form is exact is decided on the host (see
        // divisor_is_fast) for everything but the even integers, so a slider store launches nothing and keeps the
        // kernel it has; only a store that flips the verdict of the node's stage (to or from an even integer that
        // failed its check) re-plans, and that picks among kernels that already exist.
        for (const Stage &sg : e->stages)
            if (sg.type == ST_FUSED && st.node >= sg.first && st.node < sg.first + sg.count) {
                HIPCHK(e, hipSetDevice(e->device));
                if (stage_fast_div(e, sg) != sg.fast_div) replan = true;
            }
    }
    return DSPFX_OK;
}

// Reverb::refresh_seconds (reverb.rs:55-71): `num_samples = max((seconds * 48000) as usize, 128)` from the CURRENT seconds
// slider, a new ring of that length, zero-filled.  The ring length is explicit in this ABI (rivulet's capacity rounding is not
// in the reference tree): a node whose seconds slider is known (given with the node, params[1] > 0, or stored since -- a stored 0.0
// included: 128 samples) derives the new length from it -- mode bit 0 picks the page-rounded reading -- so a node fresh from the
// menu (make_buffer's ring under a 0.5 s slider, reverb.rs:44-52) jumps to 24000 samples at its first slider change like the
// reference's; without one the ring keeps its length.
uint32_t reverb_refresh_len(const Node &n) {
    return n.seconds_given ? dspfx_delay_len(n.d.params[1], n.d.mode & 1) : n.D;
}
// A NEW zero ring of D samples for node n -- Reverb::refresh_seconds (reverb.rs:55-71), which in the reference is an
// allocation of at most 192 KB and a pointer swap under the node's mutex, made on every frame a drag moves a slider
// (dsp-stuff-derive/src/lib.rs:487-497, 560-568).  Here it is a change of three scalars whatever the lengths:
//   * the ring is a table of 128-row groups and a ring of D rows uses the first ceil(D / 128) of them; groups are never freed
//     or moved by a length change (a shorter ring keeps the surplus as capacity), so blocks in flight -- which carry their own
//     D, position and clear count in their kernel arguments -- are untouched: NO wait for the device;
//   * nothing is zeroed: `zero_left = D` makes the next D frames read their taps as +0.0 (SlotArgs::zero_rows) and every row is
//     written before it is read unmasked, so groups fresh from hipMalloc (whatever they hold) are as good as zeroed ones;
//   * a LONGER ring than the node has capacity for takes the groups the storing thread allocated for it (ring_reserve: the
//     thread that moves the slider pays for the memory, like the reference's GUI thread) and appends their addresses to the
//     device table in stream order; only when none were provided (a racing chain set, dspfx_set_delay_len) does this thread
//     allocate -- without memset, without a device-wide wait;
//   * no placement probe (dspfx_tune_placement is the explicit call), no re-plan (the kernels do not depend on D; only the
//     sub-block split does: recompute_min_delay).
// On failure (out of memory) the node keeps the ring, length and position it had.
int ring_resize(dspfx_engine *e, int idx, uint32_t D, hipStream_t s) {
    Node &n = e->nodes[(size_t)idx];
    if (D < DSPFX_BUF_SIZE) D = DSPFX_BUF_SIZE;
    const size_t N = e->desc.channels;
    const size_t need = ring_groups_for(D), have = n.groups.size();
    n.group_floats = (size_t)RING_GROUP_ROWS * N;
    std::vector<float *> fresh;
    auto give_back = [&] {
        std::lock_guard<std::mutex> lk(e->pool_mu);
        if (e->ring_pool.size() <= (size_t)idx) e->ring_pool.resize((size_t)idx + 1);
        for (float *g : fresh) e->ring_pool[(size_t)idx].push_back(g);
        fresh.clear();
    };
    if (need > have) {
        {
            std::lock_guard<std::mutex> lk(e->pool_mu);
            if ((size_t)idx < e->ring_pool.size()) {
                auto &pool = e->ring_pool[(size_t)idx];
                while (fresh.size() < need - have && !pool.empty()) {
                    fresh.push_back(pool.back());
                    pool.pop_back();
                }
            }
        }
        while (fresh.size() < need - have) {
            float *g = nullptr;
            const hipError_t err = big_alloc((void **)&g, n.group_floats * sizeof(float));
            if (err != hipSuccess) {
                (void)hipGetLastError();
                give_back();
                return fail(e, DSPFX_ERR_OOM, "delay ring of %u samples: no room for %zu more groups of %zu MiB (the ring keeps its %u samples)", D,
                            need - have, (n.group_floats * sizeof(float)) >> 20, n.D);
            }
            fresh.push_back(g);
        }
    }
    float **table = n.d_groups;
    size_t cap = n.table_cap;
    if (need > cap) {             // room for every length the seconds slider can ask for (0..=1 s, page-rounded: 376 groups)
        cap = std::max<size_t>(need, 384);
        if (hipMalloc((void **)&table, cap * sizeof(float *)) != hipSuccess) {
            (void)hipGetLastError();
            give_back();
            return fail(e, DSPFX_ERR_OOM, "delay ring of %u samples: no room for its group table", D);
        }
    }
    // ---- nothing below fails
    n.groups.insert(n.groups.end(), fresh.begin(), fresh.end());
    if (table != n.d_groups) {
        if (n.d_groups) e->retired.push_back(n.d_groups);       // blocks in flight still read it
        n.d_groups = table;
        n.table_cap = cap;
        launch_table_write(n.d_groups, 0, (unsigned)n.groups.size(), n.groups.data(), s);
    } else if (n.groups.size() > have) {
        launch_table_write(n.d_groups, (unsigned)have, (unsigned)(n.groups.size() - have), n.groups.data() + have, s);
    }
    (void)hipGetLastError();
    n.D = D;
    n.d.delay_len = D;
    n.pos = 0;
    n.zero_left = D;
    n.state_bytes = (size_t)D * N * sizeof(float);   // canonical (exported) size
    {
        std::lock_guard<std::mutex> lk(e->pend_mu);
        if ((size_t)idx < e->pub_rev.size()) {
            e->pub_rev[(size_t)idx].D = D;
            e->pub_rev[(size_t)idx].have = n.groups.size();
        }
    }
    return DSPFX_OK;
}

// Capacity for a ring of want_groups groups at node `node`, allocated by the CALLING thread into the node's pool (see
// engine.h).  No api_mu: a process call in progress is not held up, and the blocks it queues are not either (hipMalloc does
// not wait for the device).  keep_free: leave that many bytes of device memory alone (0: take what is needed).  gen: the
// chain generation the caller looked at -- groups made for a chain that was replaced meanwhile are freed again.
// quiet: a best-effort caller (dspfx_chain_set's up-front reservation) -- a reservation that is not made leaves NO trace in
// dspfx_last_error; max_share: when non-zero, do not take more than free memory / max_share for it either.
int ring_reserve(dspfx_engine *e, int node, uint64_t gen, size_t want_groups, size_t keep_free, bool quiet, size_t max_share) {
    std::lock_guard<std::mutex> alloc_lk(e->alloc_mu);
    if (hipSetDevice(e->device) != hipSuccess) return quiet ? DSPFX_ERR_HIP : fail(e, DSPFX_ERR_HIP, "hipSetDevice(%d) failed", e->device);
    const size_t gbytes = (size_t)RING_GROUP_ROWS * e->desc.channels * sizeof(float);
    size_t have = 0;
    {
        std::lock_guard<std::mutex> lk(e->pend_mu);
        if (e->chain_gen != gen || node < 0 || (size_t)node >= e->pub_rev.size()) return DSPFX_OK;   // the chain was replaced: nothing to do
        have = e->pub_rev[(size_t)node].have;
    }
    {
        std::lock_guard<std::mutex> lk(e->pool_mu);
        if ((size_t)node < e->ring_pool.size()) have += e->ring_pool[(size_t)node].size();
    }
    if (have >= want_groups) return DSPFX_OK;
    const size_t short_by = want_groups - have;
    // all or nothing: a reservation that cannot be completed holds no memory
    size_t fb = 0, tb = 0;
    const bool info = hipMemGetInfo(&fb, &tb) == hipSuccess;
    if (quiet && (!info || fb < keep_free + short_by * gbytes || (max_share && short_by * gbytes > fb / max_share))) return DSPFX_ERR_OOM;
    if (info && fb < keep_free + short_by * gbytes)
        return fail(e, DSPFX_ERR_OOM, "no room for a delay ring of %zu groups of %zu MiB at node %d: %zu more needed, %zu MiB free", want_groups, gbytes >> 20,
                    node, short_by, fb >> 20);
    std::vector<float *> got;
    for (size_t k = 0; k < short_by; ++k) {
        float *g = nullptr;
        if (big_alloc((void **)&g, gbytes) != hipSuccess) {
            (void)hipGetLastError();
            for (float *x : got) (void)hipFree(x);        // (nobody ever saw them)
            if (quiet) return DSPFX_ERR_OOM;
            return fail(e, DSPFX_ERR_OOM, "no room for a delay ring of %zu groups of %zu MiB at node %d (%zu short)", want_groups, gbytes >> 20, node,
                        short_by - got.size());
        }
        got.push_back(g);
    }
    bool stale = false;
    {
        std::lock_guard<std::mutex> lk(e->pend_mu);
        stale = e->chain_gen != gen;
    }
    if (stale) {                                           // made for a chain that was replaced meanwhile
        for (float *x : got) (void)hipFree(x);
        return DSPFX_OK;
    }
    std::lock_guard<std::mutex> lk(e->pool_mu);
    if (e->ring_pool.size() <= (size_t)node) e->ring_pool.resize((size_t)node + 1);
    e->ring_pool[(size_t)node].insert(e->ring_pool[(size_t)node].end(), got.begin(), got.end());
    return DSPFX_OK;
}

void recompute_min_delay(dspfx_engine *e) {
    e->min_delay = 0xffffffffu;
    for (const Node &nd : e->nodes)
        if (nd.d.kind == DSPFX_REVERB) e->min_delay = std::min(e->min_delay, nd.D);
}

int reverb_new_ring(dspfx_engine *e, Node &n, uint32_t D, hipStream_t s) {
    if (D < DSPFX_BUF_SIZE) D = DSPFX_BUF_SIZE;
    if (D == n.D && !n.groups.empty()) {
        n.zero_left = n.D;               // the same rows go on being overwritten: only their past is forgotten
        return DSPFX_OK;
    }
    const int rc = ring_resize(e, (int)(&n - e->nodes.data()), D, s);
    recompute_min_delay(e);              // the shortest delay line bounds the sub-block length
    return rc;
}

// Apply every queued store, in order, at this block boundary (api_mu held); biquad resets are queued on `s` -- the
// stream of the block about to be launched, or the stream the state was last used on.
int drain_pending(dspfx_engine *e, hipStream_t s) {
    std::deque<dspfx_engine::Store> todo;
    {
        std::lock_guard<std::mutex> lk(e->pend_mu);
        if (e->pending.empty()) return DSPFX_OK;
        todo.swap(e->pending);
    }
    std::vector<char> biquad_reset(e->nodes.size(), 0), reverb_refresh(e->nodes.size(), 0);
    bool replan = false;
    int rc = DSPFX_OK;
    for (const auto &st : todo) {
        const int r = apply_store(e, st, biquad_reset, reverb_refresh, replan);
        if (r && !rc) rc = r;
    }
    HIPCHK(e, hipSetDevice(e->device));
    for (size_t i = 0; i < biquad_reset.size(); ++i)
        if (biquad_reset[i] && e->nodes[i].state)
            HIPCHK(e, hipMemsetAsync(e->nodes[i].state, 0, e->nodes[i].state_bytes, s));
    for (size_t i = 0; i < reverb_refresh.size(); ++i)      // once per node and boundary: every store leaves a zero ring behind
        if (reverb_refresh[i]) {
            const int r = reverb_new_ring(e, e->nodes[i], reverb_refresh_len(e->nodes[i]), s);
            if (r && !rc) rc = r;
        }
    if (replan) {
        const int r = plan(e);
        if (r && !rc) rc = r;
    }
    {
        std::lock_guard<std::mutex> lk(e->pend_mu);
        for (const auto &st : todo) {
            e->log.push_back(dspfx_param_event{st.seq, e->frames_submitted, st.node, st.param, st.value, 0});
            if (e->log.size() > 4096) e->log.pop_front();
        }
    }
    return rc;
}

}  // namespace dspfx_host


// ------------------------------------------------------------------ library

extern "C" uint32_t dspfx_abi_version(void) { return DSPFX_ABI_VERSION; }

extern "C" const char *dspfx_strerror(int status) {
    switch (status) {
    case DSPFX_OK: return "ok";
    case DSPFX_ERR_INVALID: return "invalid argument";
    case DSPFX_ERR_NO_DEVICE: return "no HIP device (this library has no CPU fallback)";
    case DSPFX_ERR_HIP: return "HIP runtime error";
    case DSPFX_ERR_OOM: return "out of device memory";
    case DSPFX_ERR_UNSUPPORTED: return "unsupported";
    case DSPFX_ERR_STATE: return "invalid engine state";
    default: return "unknown status";
    }
}

extern "C" int dspfx_device_count(void) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}

extern "C" int dspfx_node_defaults(int kind, dspfx_node_desc *d) {
    if (!d || kind < 0 || kind >= DSPFX_N_KINDS) return DSPFX_ERR_INVALID;
    memset(d, 0, sizeof *d);
    d->kind = kind;
    switch (kind) {
    case DSPFX_GAIN: d->params[0] = 1.0f; break;                       // gain.rs:21
    case DSPFX_BIQUAD:                                                  // biquad.rs:18-41
        d->params[0] = 1.0f; d->params[1] = -0.24f; d->params[2] = 0.0f;
        d->params[3] = 0.758f; d->params[4] = 0.0f; d->params[5] = 0.0f;
        break;
    case DSPFX_LOW_PASS:
    case DSPFX_HIGH_PASS: d->params[0] = 0.5f; break;                  // low_pass.rs:20
    case DSPFX_REVERB: d->params[0] = 0.5f; d->params[1] = 0.5f; d->delay_len = 128; break; // reverb.rs:29-38; 44-52: make_buffer's 128-sample ring under the 0.5 s slider
    case DSPFX_DISTORT: d->mode = DSPFX_DIST_SOFT_CLIP; break;         // distort.rs:46-50
    case DSPFX_MIX: d->params[0] = 0.5f; break;                        // mix.rs:22-28
    case DSPFX_SIGNAL_GEN: d->params[0] = 0.5f; d->params[1] = 100.0f; break;   // signal_gen.rs:41-49 (mode Sine)
    default: break;
    }
    return DSPFX_OK;
}

extern "C" uint32_t dspfx_delay_len(float seconds, int page_round) {
    // reverb.rs:58: ((seconds * 48000.0) as usize).max(128); `as` truncates and saturates
    const float s = seconds * 48000.0f;
    uint32_t d;
    if (!(s > 0.0f)) d = 0;
    else if (s >= 4294967040.0f) d = 0xffffffffu;
    else d = (uint32_t)s;
    if (d < 128) d = 128;
    if (page_round) d = (d + 1023u) / 1024u * 1024u;
    return d;
}

extern "C" float dspfx_link_divisor(uint64_t n_connected) {
    // node.rs:166,179: sequential f32 increments; saturates once 1.0 < ulp/2
    float num_frames = 0.0001f;
    for (uint64_t i = 0; i < n_connected; ++i) {
        const float next = num_frames + 1.0f;
        if (next == num_frames) break;   // further increments are no-ops
        num_frames = next;
    }
    return num_frames;
}

// ---------------------------------------------------------------- lifecycle

extern "C" int dspfx_engine_create(const dspfx_engine_desc *desc, dspfx_engine **out) {
    if (!desc || !out) return DSPFX_ERR_INVALID;
    *out = nullptr;
    if (desc->abi_version != DSPFX_ABI_VERSION) return DSPFX_ERR_INVALID;
    if (desc->channels == 0 || desc->max_frames == 0) return DSPFX_ERR_INVALID;
    if (desc->tile_channels) {
        const uint32_t W = desc->tile_channels;
        if ((W & (W - 1)) || W < 64 || desc->channels % W) return DSPFX_ERR_INVALID;
    }
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) return DSPFX_ERR_NO_DEVICE;
    if (desc->device < 0 || desc->device >= ndev) return DSPFX_ERR_INVALID;
    if (hipSetDevice(desc->device) != hipSuccess) return DSPFX_ERR_HIP;
    jit_arm_exit_guard();
    dspfx_engine *e = new dspfx_engine();
    e->env = read_env_switches();
    e->desc = *desc;
    e->device = desc->device;
    e->hop_div = dspfx_link_divisor(1);
    std::vector<const Variant *> all;
    collect_variants(all);
    for (const Variant *v : all) {
        if (v->sigs[0] != SIG_DYN) continue;
        if (v->guard) (v->mod ? e->tail_mod : e->tail) = v;
        else if (v->mod) (v->cpl == 2 ? e->dyn_mod2 : e->dyn_mod) = v;
        else if (v->f == 8 && v->cpl == 1 && v->libm) e->dyn = v;   // fallbacks handle every node kind
    }
    e->mixpart_cols = (size_t)desc->channels / 32 + 8;   // rows of first-stage partials: one per wave of a chain kernel, one per 32-channel tile of a FIR sweep
    if (hipMalloc((void **)&e->mixpart, e->mixpart_cols * desc->max_frames * sizeof(float)) != hipSuccess ||
        hipMalloc((void **)&e->mixpart_b, (size_t)128 * desc->max_frames * sizeof(float)) != hipSuccess ||
        hipMalloc((void **)&e->mixpart_b2, (size_t)128 * desc->max_frames * sizeof(float)) != hipSuccess ||
        hipMalloc((void **)&e->mt_tickets, (MIX_SLICES + 1) * sizeof(unsigned)) != hipSuccess ||
        hipMemset(e->mt_tickets, 0, (MIX_SLICES + 1) * sizeof(unsigned)) != hipSuccess) {
        dspfx_engine_destroy(e);
        return DSPFX_ERR_OOM;
    }
    plan(e);
    *out = e;
    return DSPFX_OK;
}

extern "C" void dspfx_engine_destroy(dspfx_engine *e) {
    if (!e) return;
    (void)hipSetDevice(e->device);
    (void)hipDeviceSynchronize();            // blocks still in flight read the state that is about to be freed
    // shapes of this engine still with the background compiler: skipped if not started; a compile in flight finishes into the
    // process-wide table and the disk cache (it touches nothing of the engine), and the library's exit handler waits for it --
    // a host that re-creates its engines on every graph edit (runtime.rs:319-362) is not held up here
    for (const Stage &st : e->stages) {
        if (st.async) st.async->abandoned.store(true, std::memory_order_release);
        if (st.async_mod) st.async_mod->abandoned.store(true, std::memory_order_release);
    }
    for (Node &n : e->nodes) free_node(n);
    {
        std::lock_guard<std::mutex> alloc_lk(e->alloc_mu);      // (a thread still allocating ring capacity finishes its group first)
        std::lock_guard<std::mutex> lk(e->pool_mu);
        for (auto &pool : e->ring_pool)
            for (float *g : pool) (void)hipFree(g);
        e->ring_pool.clear();
    }
    for (void *t : e->retired) (void)hipFree(t);
    e->retired.clear();
    if (e->mixpart) (void)hipFree(e->mixpart);
    if (e->mixpart_b) (void)hipFree(e->mixpart_b);
    if (e->mixpart_b2) (void)hipFree(e->mixpart_b2);
    if (e->mt_tickets) (void)hipFree(e->mt_tickets);
    for (int i = 0; i < 2; ++i) {
        if (e->mixpart2[i]) (void)hipFree(e->mixpart2[i]);
        if (e->ev_chain[i]) (void)hipEventDestroy(e->ev_chain[i]);
        if (e->ev_red[i]) (void)hipEventDestroy(e->ev_red[i]);
    }
    for (hipEvent_t ev : e->hev) (void)hipEventDestroy(ev);
    if (e->hs_in) (void)hipStreamDestroy(e->hs_in);
    if (e->hs_out) (void)hipStreamDestroy(e->hs_out);
    if (e->hs_run) (void)hipStreamDestroy(e->hs_run);
    if (e->h_in) (void)hipFree(e->h_in);
    if (e->h_side) (void)hipFree(e->h_side);
    if (e->h_out) (void)hipFree(e->h_out);
    if (e->h_mix) (void)hipFree(e->h_mix);
    for (auto &st : e->prof)
        for (auto &p : st) {
            (void)hipEventDestroy(p.first);
            (void)hipEventDestroy(p.second);
        }
    for (hipEvent_t ev : e->ev_pool) (void)hipEventDestroy(ev);
    if (e->ev_order) (void)hipEventDestroy(e->ev_order);
    delete e;
}

extern "C" const char *dspfx_last_error(const dspfx_engine *e) {
    if (!e) return "null engine";
    if (tl_err_engine != e) {           // this thread has not failed on this engine: the engine's most recent message
        std::lock_guard<std::mutex> lk(e->err_mu);
        tl_err = e->err;
        tl_err_engine = e;
    }
    return tl_err.c_str();              // valid until this thread's next failing call
}

namespace dspfx_host {
int set_nodes(dspfx_engine *e, const dspfx_node_desc *nodes, int n_nodes);
}
extern "C" int dspfx_chain_set(dspfx_engine *e, const dspfx_node_desc *nodes, int n_nodes) {
    if (!e) return DSPFX_ERR_INVALID;
    ApiScope api(e);
    e->graph_mode = false;
    e->wiring.clear();
    e->no_long = false;
    return set_nodes(e, nodes, n_nodes);
}
namespace dspfx_host {
int set_nodes(dspfx_engine *e, const dspfx_node_desc *nodes, int n_nodes) {
    if (n_nodes < 0 || n_nodes > DSPFX_MAX_NODES || (n_nodes > 0 && !nodes))
        return fail(e, DSPFX_ERR_INVALID, "chain length %d out of range", n_nodes);
    HIPCHK(e, hipSetDevice(e->device));
    e->env = read_env_switches();        // a setup call: the one place (with engine creation) where the environment is looked at
    for (int i = 0; i < n_nodes; ++i) {
        const int rc = validate_node(e, nodes[i]);
        if (rc) return rc;
    }
    {   // blocks still in flight read the state about to be freed
        const int rc = quiesce(e);
        if (rc) return rc;
    }
    for (Node &n : e->nodes) free_node(n);
    for (void *t : e->retired) (void)hipFree(t);       // group tables replaced under blocks that have finished by now
    e->retired.clear();
    e->nodes.clear();
    e->nodes.resize((size_t)n_nodes);
    e->mp_count = 0;
    if (e->mt_tickets) HIPCHK(e, hipMemset(e->mt_tickets, 0, (MIX_SLICES + 1) * sizeof(unsigned)));   // (the device is idle: quiesce above)
    int rc = DSPFX_OK;
    for (int i = 0; i < n_nodes && rc == DSPFX_OK; ++i) {
        Node &n = e->nodes[(size_t)i];
        n.d = nodes[i];
        n.d.taps = nullptr;
        if (n.d.kind == DSPFX_BIQUAD) biquad_regenerate(n);
        if (n.d.kind == DSPFX_REVERB) {
            n.D = nodes[i].delay_len;
            n.seconds_given = nodes[i].params[1] > 0.0f;
        }
        if (n.d.kind == DSPFX_FIR) n.taps.assign(nodes[i].taps, nodes[i].taps + nodes[i].n_taps);
        rc = alloc_node_state(e, n);
    }
    if (rc != DSPFX_OK) {               // leave no half-built chain behind
        for (Node &n : e->nodes) free_node(n);
        e->nodes.clear();
        (void)plan(e);
        publish_kinds(e);
        return rc;
    }
    rc = plan(e);
    publish_kinds(e);
    // A node fresh from the menu sits on make_buffer()'s short ring under a seconds slider that asks for a longer one
    // (reverb.rs:44-52): its first slider change -- any slider -- jumps to that length (reverb.rs:55-71), and the thread that
    // makes that store allocates the groups for it (enqueue_store), like the reference's GUI thread.  As a convenience the groups
    // are reserved NOW, while nothing is running, when that is cheap: by default only if they take no more than 1/16 of the
    // device's free memory and leave 8 GiB alone (at 262 144 channels the half-second ring is 23.5 GiB, at 2^20
    // channels 94 GiB -- memory a host that never touches the slider would lose to a ring it never uses, and that only
    // dspfx_ring_trim gives back).  DSPFX_MENU_RING_RESERVE=0: never; =1: whenever it fits (8 GiB left).  A host that wants
    // the O(1) first touch at any size says so itself: dspfx_reserve_delay_len.  Best effort and QUIET: a reservation that is
    // not made is not an error of dspfx_chain_set and leaves dspfx_last_error alone.
    uint64_t gen = 0;
    {
        std::lock_guard<std::mutex> lk(e->pend_mu);
        gen = e->chain_gen;
    }
    const bool never = e->env.menu_ring_reserve == 0, always = e->env.menu_ring_reserve > 0;     // (read at setup: EnvSwitches)
    for (int i = 0; i < n_nodes && rc == DSPFX_OK && !never; ++i) {
        const Node &n = e->nodes[(size_t)i];
        if (n.d.kind != DSPFX_REVERB) continue;
        const uint32_t want = reverb_refresh_len(n);
        if (ring_groups_for(want) > n.groups.size()) (void)ring_reserve(e, i, gen, ring_groups_for(want), (size_t)8 << 30, true, always ? 0 : 16);
    }
    const int rs = settle_null_stream(e);
    return rc ? rc : rs;
}
}  // namespace dspfx_host


extern "C" int dspfx_graph_set(dspfx_engine *e, const dspfx_node_desc *nodes, int n_nodes, const dspfx_graph_link *links,
                               int n_links) {
    if (!e) return DSPFX_ERR_INVALID;
    ApiScope api(e);
    const uint32_t N = e->desc.channels;
    if (N % 64u) return fail(e, DSPFX_ERR_UNSUPPORTED, "graph kernel needs whole waves: channels %% 64 == 0");
    const int vrc = validate_graph(e, nodes, n_nodes, links, n_links);
    if (vrc) return vrc;
    e->graph_mode = true;
    e->wiring.assign(links, links + n_links);
    const int rc = set_nodes(e, nodes, n_nodes);
    if (rc != DSPFX_OK) {   // leave a usable (empty) chain engine behind
        std::string msg;
        {
            std::lock_guard<std::mutex> lk(e->err_mu);      // (a GUI thread's failing store may be writing it)
            msg = e->err;
        }
        (void)dspfx_chain_set(e, nullptr, 0);
        std::lock_guard<std::mutex> lk(e->err_mu);
        e->err = msg;
    }
    return rc;
}

// Kernels compiled in the background are adopted at a block boundary; a host (or a benchmark) that wants them BEFORE its
// first block waits here.
extern "C" int dspfx_kernels_ready(dspfx_engine *e, int wait_ms) {
    if (!e) return DSPFX_ERR_INVALID;
    const auto t0 = std::chrono::steady_clock::now();
    for (;;) {
        std::vector<std::shared_ptr<AsyncJit>> jobs;
        {
            ApiScope api(e);
            if (api.rc) return api.rc;
            bool pending = false;
            for (const Stage &st : e->stages) {
                if (st.type != ST_FUSED) continue;
                if (st.async || st.async_mod) adopt_async_jit(e, st);       // whatever is finished takes effect now (the engine is ours)
                if (st.async) jobs.push_back(st.async);
                if (st.async_mod) jobs.push_back(st.async_mod);
                pending = pending || st.async || st.async_mod;
            }
            if (!pending) return 1;
        }
        // wait WITHOUT the engine's lock: the thread that drives the blocks is not held up by this one
        const int left = wait_ms - (int)std::chrono::duration_cast<std::chrono::milliseconds>(std::chrono::steady_clock::now() - t0).count();
        if (left <= 0) return 0;
        bool all = true;
        for (const auto &j : jobs) all = async_jit_wait(j, std::max(1, left)) && all;
        if (!all) return 0;
    }
}

extern "C" int dspfx_chain_len(const dspfx_engine *e) {
    if (!e) return DSPFX_ERR_INVALID;
    std::lock_guard<std::recursive_mutex> lk(e->api_mu);
    return (int)e->nodes.size();
}

// Slider / mode stores.  The reference's GUI thread stores into an atomic while the node's task runs process()
// (dsp-stuff-derive/src/lib.rs:487-492) and runs after_settings_change under the node's mutex (biquad.rs:62-76): the
// store lands between two blocks.  Here: the store is queued (never waiting for a process call in progress) and applied
// by the next entry point that holds the engine -- at once when the engine is idle, else at the next block boundary --
// with its state write (the biquad reset) queued in stream order behind the blocks already in flight.
namespace dspfx_host {
int enqueue_store(dspfx_engine *e, int node, int param, float value, uint64_t *seq_out) {
    size_t want_groups = 0;
    uint64_t gen = 0;
    {
        std::lock_guard<std::mutex> lk(e->pend_mu);
        if (node < 0 || node >= (int)e->pub_kinds.size() || param < -1 || param >= 8)
            return fail(e, DSPFX_ERR_INVALID, param < 0 ? "node %d out of range" : "set_param(%d,%d) out of range", node, param);
        if (param < 0) {                                // a mode store: validate against the node's kind
            dspfx_node_desc d{};
            d.kind = e->pub_kinds[(size_t)node];
            d.mode = (int)value;
            d.delay_len = DSPFX_BUF_SIZE;
            d.n_taps = 1;
            static const double one = 1.0;
            d.taps = &one;
            const int rc = validate_node(e, d);
            if (rc) return rc;
        }
        if (e->pub_kinds[(size_t)node] == DSPFX_REVERB && param >= 0) {
            // the seconds slider is 0..=1 (reverb.rs:34-37); anything else is not a value the reference's widget can hold --
            // and would ask for a ring of up to 2^32 samples per channel
            if (param == 1 && !(value >= 0.0f && value <= 1.0f))
                return fail(e, DSPFX_ERR_INVALID, "REVERB seconds %g outside the slider's range 0..=1 (reverb.rs:34-37)", (double)value);
            // the ring this store will swap in (reverb_refresh_len, with the stores queued before it applied)
            const dspfx_engine::PubReverb &pr = e->pub_rev[(size_t)node];
            const float seconds = param == 1 ? value : pr.seconds;
            if (param == 1 || pr.given) want_groups = ring_groups_for(dspfx_delay_len(seconds, pr.mode & 1));
            if (want_groups <= pr.have) want_groups = 0;
        }
        gen = e->chain_gen;                             // every kind: what was validated above was THIS chain's node
    }
    // Reverb::refresh_seconds allocates the new ring on the thread that moved the slider (reverb.rs:55-71).  So does this: a
    // ring longer than the node's capacity gets its groups HERE, before the store is queued -- the thread that drives the blocks
    // finds them at the block boundary and only swaps.  Out of memory: the store is not made, the node keeps ring and slider.
    if (want_groups) {
        const int rc = ring_reserve(e, node, gen, want_groups, 0, false, 0);
        if (rc) return rc;
    }
    {
        std::lock_guard<std::mutex> lk(e->pend_mu);
        // dspfx_chain_set from another thread between the two locked sections (the lock is dropped for the allocation): the kind,
        // range and ring length were checked against a node that is gone -- the store is NOT made
        if (e->chain_gen != gen || node >= (int)e->pub_kinds.size())
            return fail(e, DSPFX_ERR_STATE, "the chain was replaced while the store to node %d was being made: not stored", node);
        if (e->pub_kinds[(size_t)node] == DSPFX_REVERB) {
            if (param == 1) {
                e->pub_rev[(size_t)node].seconds = value;
                e->pub_rev[(size_t)node].given = true;
            }
            if (param < 0) e->pub_rev[(size_t)node].mode = (int)value;
        }
        const uint64_t seq = e->next_seq++;
        e->pending.push_back(dspfx_engine::Store{seq, node, param, value});
        if (seq_out) *seq_out = seq;
    }
    if (e->api_mu.try_lock()) {                         // nobody is inside the engine: the store takes effect now
        std::lock_guard<std::recursive_mutex> lk(e->api_mu, std::adopt_lock);
        if (e->api_depth == 0) {                        // (not this thread re-entering from inside an entry point)
            ++e->api_depth;
            int rc = hipSetDevice(e->device) == hipSuccess ? DSPFX_OK : fail(e, DSPFX_ERR_HIP, "hipSetDevice(%d) failed", e->device);
            if (rc == DSPFX_OK) rc = drain_pending(e, e->cur_stream_set ? e->cur_stream : nullptr);
            --e->api_depth;
            return rc;
        }
    }
    return DSPFX_OK;                                    // applied at the next block boundary
}
}  // namespace dspfx_host

extern "C" int dspfx_set_param(dspfx_engine *e, int node, int param, float value) {
    if (!e) return DSPFX_ERR_INVALID;
    if (param < 0) return fail(e, DSPFX_ERR_INVALID, "set_param(%d,%d) out of range", node, param);
    return enqueue_store(e, node, param, value, nullptr);
}

extern "C" int dspfx_set_param_seq(dspfx_engine *e, int node, int param, float value, uint64_t *seq) {
    if (!e) return DSPFX_ERR_INVALID;
    if (param < 0) return fail(e, DSPFX_ERR_INVALID, "set_param(%d,%d) out of range", node, param);
    return enqueue_store(e, node, param, value, seq);
}

extern "C" int dspfx_set_mode(dspfx_engine *e, int node, int mode) {
    if (!e) return DSPFX_ERR_INVALID;
    return enqueue_store(e, node, -1, (float)mode, nullptr);
}

extern "C" int dspfx_param_log(dspfx_engine *e, dspfx_param_event *dst, int cap, uint64_t after_seq) {
    if (!e || cap < 0 || (cap > 0 && !dst)) return DSPFX_ERR_INVALID;
    std::lock_guard<std::mutex> lk(e->pend_mu);
    int n = 0;
    for (const dspfx_param_event &ev : e->log)
        if (ev.seq > after_seq && n < cap) dst[n++] = ev;
    return n;
}

extern "C" uint64_t dspfx_frames_submitted(const dspfx_engine *e) {
    if (!e) return 0;
    std::lock_guard<std::recursive_mutex> lk(e->api_mu);
    return e->frames_submitted;
}

extern "C" int dspfx_set_delay_len(dspfx_engine *e, int node, uint32_t delay_len) {
    if (!e) return DSPFX_ERR_INVALID;
    ApiScope api(e);
    if (api.rc) return api.rc;
    if (node < 0 || node >= (int)e->nodes.size() || e->nodes[(size_t)node].d.kind != DSPFX_REVERB)
        return fail(e, DSPFX_ERR_INVALID, "node %d is not a REVERB node", node);
    if (delay_len < DSPFX_BUF_SIZE) return fail(e, DSPFX_ERR_INVALID, "delay_len %u < 128", delay_len);
    // reverb.rs:55-71: a brand-new zero ring -- three scalars change (reverb_new_ring); a ring longer than the node's capacity
    // has its missing groups allocated here, by the calling thread, without a wait for the device (dspfx_reserve_delay_len from
    // another thread beforehand keeps even that off the thread that drives the blocks).
    return reverb_new_ring(e, e->nodes[(size_t)node], delay_len, e->cur_stream_set ? e->cur_stream : nullptr);
}

extern "C" int dspfx_reserve_delay_len(dspfx_engine *e, int node, uint32_t delay_len) {
    if (!e) return DSPFX_ERR_INVALID;
    uint64_t gen = 0;
    {
        std::lock_guard<std::mutex> lk(e->pend_mu);
        if (node < 0 || node >= (int)e->pub_kinds.size() || e->pub_kinds[(size_t)node] != DSPFX_REVERB)
            return fail(e, DSPFX_ERR_INVALID, "node %d is not a REVERB node", node);
        gen = e->chain_gen;
    }
    if (delay_len < DSPFX_BUF_SIZE) return fail(e, DSPFX_ERR_INVALID, "delay_len %u < 128", delay_len);
    return ring_reserve(e, node, gen, ring_groups_for(delay_len), 0, false, 0);
}

extern "C" int dspfx_ring_trim(dspfx_engine *e) {
    if (!e) return DSPFX_ERR_INVALID;
    ApiScope api(e);
    if (api.rc) return api.rc;
    const int rc = quiesce(e);              // blocks in flight may still be on a longer ring of the same groups
    if (rc) return rc;
    std::vector<float *> drop;
    {
        std::lock_guard<std::mutex> lk(e->pool_mu);
        for (auto &pool : e->ring_pool) {
            drop.insert(drop.end(), pool.begin(), pool.end());
            pool.clear();
        }
    }
    for (size_t i = 0; i < e->nodes.size(); ++i) {
        Node &n = e->nodes[i];
        if (n.d.kind != DSPFX_REVERB) continue;
        const size_t need = ring_groups_for(n.D);
        while (n.groups.size() > need) {
            drop.push_back(n.groups.back());
            n.groups.pop_back();
        }
        std::lock_guard<std::mutex> lk(e->pend_mu);
        if (i < e->pub_rev.size()) e->pub_rev[i].have = n.groups.size();
    }
    for (float *g : drop) (void)hipFree(g);
    for (void *t : e->retired) (void)hipFree(t);
    e->retired.clear();
    return DSPFX_OK;
}

extern "C" int dspfx_set_taps(dspfx_engine *e, int node, const double *taps_reversed, uint32_t n_taps, int mode) {
    if (!e) return DSPFX_ERR_INVALID;
    ApiScope api(e);
    if (api.rc) return api.rc;
    if (node < 0 || node >= (int)e->nodes.size() || e->nodes[(size_t)node].d.kind != DSPFX_FIR)
        return fail(e, DSPFX_ERR_INVALID, "node %d is not a FIR node", node);
    if (!taps_reversed || n_taps == 0) return fail(e, DSPFX_ERR_INVALID, "FIR node needs taps");
    {   // blocks in flight still read the old tap tables (and the ring, should it have to grow)
        const int rc = quiesce(e);
        if (rc) return rc;
    }
    Node &n = e->nodes[(size_t)node];
    n.taps.assign(taps_reversed, taps_reversed + n_taps);
    n.d.n_taps = n_taps;
    n.d.mode = mode;
    int rc;
    if (!n.fir.ring) rc = alloc_node_state(e, n);
    else {
        // fir.rs:153-171 replaces `taps` only: `state` (fir.rs:64-65) is never cleared, so the history survives
        rc = fir_set_taps(n.fir, n.taps.data(), n_taps, mode);
        if (rc) rc = fail(e, rc, "FIR tap reload failed: %s", fir_last_error());
    }
    const int rs = settle_null_stream(e);
    return rc ? rc : rs;
}

extern "C" int dspfx_set_fir_precision(dspfx_engine *e, int node, int precision) {
    if (!e) return DSPFX_ERR_INVALID;
    ApiScope api(e);
    if (api.rc) return api.rc;
    if (node < 0 || node >= (int)e->nodes.size() || e->nodes[(size_t)node].d.kind != DSPFX_FIR)
        return fail(e, DSPFX_ERR_INVALID, "node %d is not a FIR node", node);
    if (precision < DSPFX_FIR_PRECISION_DEFAULT || precision > DSPFX_FIR_PRECISION_HALF)
        return fail(e, DSPFX_ERR_INVALID, "unknown FIR precision %d", precision);
    e->nodes[(size_t)node].fir.precision = precision;
    return DSPFX_OK;
}

namespace dspfx_host {
// Zero every node's DSP state, queued on `s`: ordered behind the blocks in flight there, ahead of the next one.
int reset_on(dspfx_engine *e, hipStream_t s) {
    for (Node &n : e->nodes) {
        if (n.state) HIPCHK(e, hipMemsetAsync(n.state, 0, n.d.kind == DSPFX_BIQUAD ? 4 * (size_t)e->desc.channels * sizeof(float)
                                                                                   : (size_t)e->desc.channels * sizeof(float), s));
        if (!n.groups.empty()) n.zero_left = n.D;   // a zero ring without touching its (up to 94 GiB of) rows: see Node::zero_left
        if (n.d.kind == DSPFX_FIR) fir_reset(n.fir, s);
    }
    e->mp_count = 0;   // blocks still in the mix pipeline are dropped
    // the in-launch bus' arrival counters are zero between launches only if every launch of a block ran to its end: put them
    // back here (stream-ordered), so that a launch that failed half-way cannot leave every later bus stale
    if (e->mt_tickets) HIPCHK(e, hipMemsetAsync(e->mt_tickets, 0, (MIX_SLICES + 1) * sizeof(unsigned), s));
    return DSPFX_OK;
}
}  // namespace dspfx_host

extern "C" int dspfx_reset(dspfx_engine *e) {
    if (!e) return DSPFX_ERR_INVALID;
    ApiScope api(e);
    if (api.rc) return api.rc;
    return reset_on(e, e->cur_stream_set ? e->cur_stream : nullptr);
}

// ------------------------------------------------------------- the hot path

// A few-channel engine (its fused stages have time-sliced kernels, which take blocks of exactly 128 frames) given a longer
// block that is whole 128-frame blocks: 128 if the call should go through them block by block, else 0.  One 256-frame launch
// of the standard kernel at 4096 / 16384 / 32768 channels of the 3-node chain: 46.8 / 51.8 / 58.3 us; two time-sliced
// launches: 21.2 / 22.7 / 28.5 (5-node chain: 52.0 / 53.7 / 55.7 against 31.7 / 34.1 / 44.6; profiles/r03_small_n.txt).
// B = 256 IS two reference blocks back to back, so nothing changes but the launches.
static uint32_t ts_sub_block(const dspfx_engine *e) {
    if (e->win_n || e->mp_building || e->partials_override) return 0;     // channel windows, the pipelined / deferred bus: one launch per call
    uint32_t b = 0;
    for (const Stage &st : e->stages)
        if (st.type == ST_FUSED && st.count > 0) {
            if (!st.var_ts) return 0;
            b = 4u * (uint32_t)st.var_ts->ts;
        }
    for (const Node &nd : e->nodes)
        if (nd.latch_valid || nd.ctl_now[0] || nd.ctl_now[1] || nd.ctl_now[2]) return 0;     // control ports: the interpreter serves
    return b;
}

extern "C" int dspfx_process(dspfx_engine *e, const float *in, const float *side, float *out, float *mix,
                             uint32_t n_frames, void *stream) {
    if (!e) return DSPFX_ERR_INVALID;
    hipStream_t s = (hipStream_t)stream;
    ApiScope api(e, true, s);            // queued slider stores take effect here, at the block boundary
    if (api.rc) return api.rc;
    if (!in || !out) return fail(e, DSPFX_ERR_INVALID, "in/out must not be null");
    if (n_frames == 0) return DSPFX_OK;
    if (n_frames > e->desc.max_frames)
        return fail(e, DSPFX_ERR_INVALID, "n_frames %u > max_frames %u", n_frames, e->desc.max_frames);
    if (e->has_fuzz && n_frames % DSPFX_BUF_SIZE)
        return fail(e, DSPFX_ERR_INVALID, "Fuzz is block-global over 128 frames: n_frames %u %% 128 != 0", n_frames);
    // a block's delay taps must not depend on the same launch's outputs: split at min delay
    uint32_t sub = std::min(n_frames, e->min_delay);
    if (e->has_fuzz || e->has_siggen) sub = std::max<uint32_t>(DSPFX_BUF_SIZE, sub / DSPFX_BUF_SIZE * DSPFX_BUF_SIZE);
    if (const uint32_t tsb = ts_sub_block(e))
        if (n_frames > tsb && n_frames % tsb == 0 && sub >= tsb && sub % tsb == 0) sub = tsb;
    const size_t N = e->desc.channels;
    for (uint32_t f0 = 0; f0 < n_frames; f0 += sub) {
        const uint32_t nf = std::min(sub, n_frames - f0);
        // frame f0 of a block starts f0 rows in: a row is N floats (frame-major) or W floats (tiled)
        const size_t off = (size_t)f0 * (e->desc.tile_channels ? e->desc.tile_channels : N);
        e->io_off = off;
        const int rc = run_subblock(e, in + off, side ? side + off : nullptr, out + off, mix ? mix + f0 : nullptr,
                                    nf, e->ctl_tile_frames ? e->ctl_tile_frames : n_frames, s);
        e->io_off = 0;
        if (rc) return rc;
        e->frames_submitted += nf;
    }
    return DSPFX_OK;
}

// The Output node in the same launch: nodes/output.rs:215-249 feeds every channel's pipe into one port, and
// collect_and_average (node.rs:162-194) sums them and divides by f32(0.0001 + n).
extern "C" int dspfx_process_bus(dspfx_engine *e, const float *in, const float *side, float *out, float *mix,
                                 uint32_t n_frames, uint64_t n_connected, void *stream) {
    if (!e) return DSPFX_ERR_INVALID;
    ApiScope api(e, true, (hipStream_t)stream);
    if (api.rc) return api.rc;
    if (!mix) return fail(e, DSPFX_ERR_INVALID, "mix must not be null");
    float div = 0.0f;
    if (n_connected) {
        if (e->div_n != n_connected || e->div_v == 0.0f) {
            e->div_v = dspfx_link_divisor(n_connected);
            e->div_n = n_connected;
        }
        div = e->div_v;
    }
    e->bus_div_now = div;
    const int rc = dspfx_process(e, in, side, out, mix, n_frames, stream);
    e->bus_div_now = 0.0f;
    return rc;
}

extern "C" int dspfx_process_io(dspfx_engine *e, const float *const *ins, int n_ins, float *const *outs, int n_outs, float *mix,
                                uint32_t n_frames, void *stream) {
    if (!e) return DSPFX_ERR_INVALID;
    ApiScope api(e, true, (hipStream_t)stream);
    if (api.rc) return api.rc;
    if (n_ins < 0 || n_ins > DSPFX_GRAPH_MAX_IO || n_outs < 1 || n_outs > DSPFX_GRAPH_MAX_IO || (n_ins > 0 && !ins) || !outs || !outs[0])
        return fail(e, DSPFX_ERR_INVALID, "process_io: 0..%d input blocks, 1..%d output blocks (the first one non-null)", DSPFX_GRAPH_MAX_IO, DSPFX_GRAPH_MAX_IO);
    if (!e->graph_mode && (n_ins > 2 || n_outs > 1)) return fail(e, DSPFX_ERR_INVALID, "only a graph engine (dspfx_graph_set) takes extra blocks");
    for (int k = 0; k < GRAPH_IO; ++k) {
        e->io_in[k] = k < n_ins ? ins[k] : nullptr;
        e->io_out[k] = k < n_outs ? outs[k] : nullptr;
    }
    // a graph that never reads `in` still gets a valid pointer (the kernel does not touch it)
    const float *in0 = (n_ins > 0 && ins[0]) ? ins[0] : outs[0];
    const int rc = dspfx_process(e, in0, n_ins > 1 ? ins[1] : nullptr, outs[0], mix, n_frames, stream);
    for (int k = 0; k < GRAPH_IO; ++k) {
        e->io_in[k] = nullptr;
        e->io_out[k] = nullptr;
    }
    return rc;
}

extern "C" int dspfx_process_ctl(dspfx_engine *e, const float *in, const float *side, float *out, float *mix,
                                 uint32_t n_frames, const dspfx_ctl *ctl, int n_ctl, void *stream) {
    if (!e) return DSPFX_ERR_INVALID;
    ApiScope api(e, true, (hipStream_t)stream);
    if (api.rc) return api.rc;
    if (n_ctl < 0 || (n_ctl > 0 && !ctl)) return fail(e, DSPFX_ERR_INVALID, "bad control-port list");
    if (n_ctl > 0 && e->graph_mode) return fail(e, DSPFX_ERR_INVALID, "a fused graph's control ports are links of the graph");
    if (n_ctl > 0 && !e->no_long) {   // control ports are evaluated by the chain kernels: cut long stages back to their size
        bool has_long = false;
        for (const Stage &st : e->stages) has_long = has_long || (st.type == ST_FUSED && st.count > MAX_SLOTS);
        if (has_long) {
            e->no_long = true;
            const int rc = plan(e);
            if (rc) return rc;
        }
    }
    for (int i = 0; i < n_ctl; ++i) {
        const dspfx_ctl &c = ctl[i];
        if (c.node < 0 || c.node >= (int)e->nodes.size() || !c.signal)
            return fail(e, DSPFX_ERR_INVALID, "control port %d: bad node or null signal", i);
        Node &n = e->nodes[(size_t)c.node];
        int n_sliders = 0;
        switch (n.d.kind) {
        case DSPFX_GAIN: n_sliders = 1; break;
        case DSPFX_DISTORT: n_sliders = 1; break;   // Fuzz too: distort.rs:176-180 maps the level port before the mode switch
        case DSPFX_OVERDRIVE: n_sliders = 3; break;
        case DSPFX_MIX: n_sliders = 1; break;
        case DSPFX_SIGNAL_GEN: n_sliders = 2; break;
        default: break;
        }
        if (c.param < 0 || c.param >= n_sliders)
            return fail(e, DSPFX_ERR_INVALID, "node %d has no `as_input` slider %d", c.node, c.param);
        if (!n.latch[c.param]) HIPCHK(e, hipMalloc((void **)&n.latch[c.param], (size_t)e->desc.channels * sizeof(float)));
    }
    // split like dspfx_process does, offsetting the control signals with the samples
    const size_t N = e->desc.channels;
    const size_t rowlen = e->desc.tile_channels ? e->desc.tile_channels : N;
    if (n_frames > e->desc.max_frames) return fail(e, DSPFX_ERR_INVALID, "n_frames %u > max_frames %u", n_frames, e->desc.max_frames);
    uint32_t sub = std::min(n_frames, e->min_delay);
    if (e->has_fuzz || e->has_siggen) sub = std::max<uint32_t>(DSPFX_BUF_SIZE, sub / DSPFX_BUF_SIZE * DSPFX_BUF_SIZE);
    int rc = DSPFX_OK;
    for (uint32_t f0 = 0; f0 < n_frames && rc == DSPFX_OK; f0 += sub) {
        const uint32_t nf = std::min(sub, n_frames - f0);
        const size_t off = (size_t)f0 * rowlen;
        for (int i = 0; i < n_ctl; ++i) e->nodes[(size_t)ctl[i].node].ctl_now[ctl[i].param] = ctl[i].signal + off;
        // one sub-block == one dspfx_process call of nf frames on offset pointers (tile stride stays n_frames)
        e->ctl_tile_frames = n_frames;
        rc = dspfx_process(e, in + off, side ? side + off : nullptr, out + off, mix ? mix + f0 : nullptr, nf, stream);
        e->ctl_tile_frames = 0;
    }
    for (Node &n : e->nodes)
        for (int k = 0; k < 3; ++k) n.ctl_now[k] = nullptr;
    return rc;
}

extern "C" int dspfx_process_partials(dspfx_engine *e, const float *in, const float *side, float *out,
                                      uint32_t n_frames, void *stream) {
    if (!e) return DSPFX_ERR_INVALID;
    hipStream_t s = (hipStream_t)stream;
    ApiScope api(e, true, s);
    if (api.rc) return api.rc;
    if (e->collect_due) return fail(e, DSPFX_ERR_STATE, "dspfx_mix_collect must follow dspfx_process_partials");
    if (n_frames == 0 || n_frames > e->desc.max_frames) return fail(e, DSPFX_ERR_INVALID, "bad n_frames %u", n_frames);
    if (n_frames > e->min_delay)
        return fail(e, DSPFX_ERR_UNSUPPORTED, "deferred mix needs n_frames <= shortest delay line (%u)", e->min_delay);
    const int b = e->flip;
    if (!e->mixpart2[b]) {
        HIPCHK(e, hipMalloc((void **)&e->mixpart2[b], e->mixpart_cols * e->desc.max_frames * sizeof(float)));
        HIPCHK(e, hipEventCreateWithFlags(&e->ev_chain[b], hipEventDisableTiming));
        HIPCHK(e, hipEventCreateWithFlags(&e->ev_red[b], hipEventDisableTiming));
    }
    if (e->red_pending[b]) {   // the collect that read this buffer two blocks ago must be done
        if (hipEventQuery(e->ev_red[b]) != hipSuccess)      // normally long finished: no packet needed
            HIPCHK(e, hipStreamWaitEvent(s, e->ev_red[b], 0));
        e->red_pending[b] = false;
    }
    e->partials_override = e->mixpart2[b];
    const int rc = dspfx_process(e, in, side, out, nullptr, n_frames, stream);
    e->partials_override = nullptr;
    if (rc) return rc;
    HIPCHK(e, hipEventRecord(e->ev_chain[b], s));
    e->collect_due = true;
    return DSPFX_OK;
}

extern "C" int dspfx_mix_collect(dspfx_engine *e, float *mix, uint32_t n_frames, void *stream) {
    if (!e || !mix) return DSPFX_ERR_INVALID;
    ApiScope api(e);                     // (its stream only reads the partial sums: the DSP state stays bound to the chain's stream)
    if (api.rc) return api.rc;
    if (!e->collect_due) return fail(e, DSPFX_ERR_STATE, "no partials pending: call dspfx_process_partials first");
    const int b = e->flip;
    if (n_frames != e->part_frames[b]) return fail(e, DSPFX_ERR_INVALID, "n_frames %u != %u of the pending block", n_frames, e->part_frames[b]);
    HIPCHK(e, hipSetDevice(e->device));
    hipStream_t s = (hipStream_t)stream;
    HIPCHK(e, hipStreamWaitEvent(s, e->ev_chain[b], 0));
    launch_mix_reduce(e->mixpart2[b], e->mixpart_b2, mix, n_frames, e->part_stride[b], s);
    HIPCHK(e, hipGetLastError());
    HIPCHK(e, hipEventRecord(e->ev_red[b], s));
    e->red_pending[b] = true;
    e->collect_due = false;
    e->flip ^= 1;
    return DSPFX_OK;
}

extern "C" int dspfx_process_mixpipe(dspfx_engine *e, const float *in, const float *side, float *out, float *mix,
                                     uint32_t n_frames, uint64_t n_connected, void *stream) {
    if (!e) return DSPFX_ERR_INVALID;
    ApiScope api(e, true, (hipStream_t)stream);
    if (api.rc) return api.rc;
    if (e->collect_due) return fail(e, DSPFX_ERR_STATE, "dspfx_mix_collect must follow dspfx_process_partials");
    if (n_frames == 0 || n_frames > e->desc.max_frames) return fail(e, DSPFX_ERR_INVALID, "bad n_frames %u", n_frames);
    if (n_frames > e->min_delay)
        return fail(e, DSPFX_ERR_UNSUPPORTED, "pipelined mix needs n_frames <= shortest delay line (%u)", e->min_delay);
    if (e->mp_count && n_frames != e->mp_frames)
        return fail(e, DSPFX_ERR_STATE, "n_frames changed from %u to %u inside the mix pipeline: flush first", e->mp_frames, n_frames);
    if (e->mp_count >= 2 && !mix) return fail(e, DSPFX_ERR_INVALID, "mix must not be null from the third block on");
    HIPCHK(e, hipSetDevice(e->device));
    for (int b = 0; b < 2; ++b)
        if (!e->mixpart2[b]) {
            HIPCHK(e, hipMalloc((void **)&e->mixpart2[b], e->mixpart_cols * e->desc.max_frames * sizeof(float)));
            HIPCHK(e, hipEventCreateWithFlags(&e->ev_chain[b], hipEventDisableTiming));
            HIPCHK(e, hipEventCreateWithFlags(&e->ev_red[b], hipEventDisableTiming));
        }
    float div = 0.0f;
    if (n_connected) {
        if (e->div_n != n_connected || e->div_v == 0.0f) {
            e->div_v = dspfx_link_divisor(n_connected);
            e->div_n = n_connected;
        }
        div = e->div_v;
    }
    e->mp_frames = n_frames;
    e->mp_mix_now = mix;
    e->mp_div_now = div;
    e->mp_building = true;
    const int rc = dspfx_process(e, in, side, out, nullptr, n_frames, stream);
    e->mp_building = false;
    if (rc) return rc;
    ++e->mp_count;
    return DSPFX_OK;
}

extern "C" int dspfx_mixpipe_flush(dspfx_engine *e, float *mix_older, float *mix_newer, uint64_t n_connected, void *stream) {
    if (!e) return DSPFX_ERR_INVALID;
    ApiScope api(e, true, (hipStream_t)stream);
    if (api.rc) return api.rc;
    if (e->mp_count == 0) return DSPFX_OK;
    if (!mix_newer || (e->mp_count >= 2 && !mix_older)) return fail(e, DSPFX_ERR_INVALID, "mix buffers must not be null");
    HIPCHK(e, hipSetDevice(e->device));
    hipStream_t s = (hipStream_t)stream;
    const uint32_t nf = e->mp_frames;
    float div = 0.0f;
    if (n_connected) {
        if (e->div_n != n_connected || e->div_v == 0.0f) {
            e->div_v = dspfx_link_divisor(n_connected);
            e->div_n = n_connected;
        }
        div = e->div_v;
    }
    const int last = (int)((e->mp_count - 1) & 1);          // buffers written by the last launch
    float *b_last = last ? e->mixpart_b2 : e->mixpart_b, *b_other = last ? e->mixpart_b : e->mixpart_b2;
    if (e->mp_count >= 2) {                                  // block n-2: its slices were reduced by the last launch
        launch_mix_reduce_final(b_last, mix_older, nf, s);
        if (div != 0.0f) launch_mix_finish(mix_older, nf, div, s);
    }
    launch_mix_reduce_slices(e->mixpart2[last], b_other, nf, e->mp_rows[last], s);   // block n-1: only its partials exist
    launch_mix_reduce_final(b_other, mix_newer, nf, s);
    if (div != 0.0f) launch_mix_finish(mix_newer, nf, div, s);
    HIPCHK(e, hipGetLastError());
    e->mp_count = 0;
    return DSPFX_OK;
}

extern "C" int dspfx_mix_finish(dspfx_engine *e, float *mix, uint32_t n_frames, uint64_t n_connected, void *stream) {
    if (!e || !mix) return DSPFX_ERR_INVALID;
    ApiScope api(e);
    if (api.rc) return api.rc;
    HIPCHK(e, hipSetDevice(e->device));
    if (e->div_n != n_connected || e->div_v == 0.0f) {   // the literal f32 increment loop is O(n): cache it
        e->div_v = dspfx_link_divisor(n_connected);
        e->div_n = n_connected;
    }
    launch_mix_finish(mix, n_frames, e->div_v, (hipStream_t)stream);
    HIPCHK(e, hipGetLastError());
    return DSPFX_OK;
}

