// plan.hip -- which kernels serve a chain: the exact-division decision, variant selection (compiled in, specialised at run
// time, time-sliced, the guarded form for channels left over), the background specialisation of small engines, and plan(),
// which cuts the chain into stages.  See engine.h for the split.
#include "engine.h"

using namespace dspfx;
using namespace dspfx_host;

namespace dspfx_host {


// Is (float)((double)x * RN_f64(1/c)) == x / c for every f32 x?  (div_c in chain_kernels.hip.h has the argument.)
//   * c not an even integer, or a power of two: yes, by the theorem there -- decided here, no device involved, so a
//     slider store, a fan-in divisor f32(0.0001 + k) or dspfx_graph_source never launch anything;
//   * c an even integer that is not a power of two (6, 10, 12, ... -- exact ties exist among its subnormal quotients):
//     the exhaustive 2^32-input check decides, on the CURRENT device (plan() selects the engine's device first).  Only
//     COMPLETED checks are cached: a HIP failure answers "not fast" for this call and is asked again next time;
//     have_device = false only consults the cache.
// DSPFX_FAST_DIV=0 forces the IEEE path (A/B runs; part of the engine's environment snapshot: forced_off).
bool divisor_is_fast(float c, bool have_device, bool forced_off) {
    static std::mutex mu;
    static std::map<uint32_t, bool> cache;
    if (forced_off) return false;
    if (!(c == c) || c == 0.0f || std::isinf(c)) return false;
    const float ac = fabsf(c), half = ac * 0.5f;
    int ex = 0;
    const bool even_integer = ac >= 2.0f && half == floorf(half);
    const bool pow2 = frexpf(ac, &ex) == 0.5f;
    if (!even_integer || pow2) return true;
    uint32_t bits;
    memcpy(&bits, &c, 4);
    std::lock_guard<std::mutex> lk(mu);
    auto it = cache.find(bits);
    if (it != cache.end()) return it->second;
    if (!have_device) return false;
    bool done = false, ok = false;
    unsigned long long *d = nullptr, h = 1;
    if (hipMalloc((void **)&d, sizeof h) == hipSuccess) {
        if (hipMemset(d, 0, sizeof h) == hipSuccess && verify_divisor_on_device(c, 1.0 / (double)c, d, nullptr) == 0 &&
            hipMemcpy(&h, d, sizeof h, hipMemcpyDeviceToHost) == hipSuccess) {
            done = true;
            ok = h == 0;
        }
        (void)hipFree(d);
    }
    if (!done) {
        (void)hipGetLastError();
        return false;
    }
    cache[bits] = ok;
    return ok;
}

bool node_divisors_fast(const Node &n, bool have_device, bool forced_off) {
    if (n.d.kind == DSPFX_DISTORT && (n.d.mode == DSPFX_DIST_HARD_CLIP || n.d.mode == DSPFX_DIST_SOFT_CLIP)) {
        if (n.d.params[0] < 0.001f) return true;   // bypassed: never divides
        if (!divisor_is_fast(n.d.params[0], have_device, forced_off)) return false;
        if (n.d.mode == DSPFX_DIST_SOFT_CLIP && !divisor_is_fast(3.0f, have_device, forced_off)) return false;
    }
    return true;
}

bool fusable(const Node &n) {
    if (n.d.kind == DSPFX_FIR) return false;
    if (n.d.kind == DSPFX_DISTORT && n.d.mode == DSPFX_DIST_FUZZ) return false;
    return true;
}

int node_hop(const dspfx_engine *e, int idx) {
    return idx == 0 ? ((e->desc.link_flags & DSPFX_LINK_INPUT) ? 1 : 0)
                    : ((e->desc.link_flags & DSPFX_LINK_INTERNAL) ? 1 : 0);
}

void collect_variants(std::vector<const Variant *> &out) {
    int n = 0;
    const Variant *v = variants_dyn(&n);
    for (int i = 0; i < n; ++i) out.push_back(v + i);
    v = variants_static3(&n);
    for (int i = 0; i < n; ++i) out.push_back(v + i);
    v = variants_static5(&n);
    for (int i = 0; i < n; ++i) out.push_back(v + i);
}

static int env_int(const char *name) {
    const char *v = getenv(name);
    return v ? atoi(v) : -1;
}
// The ONE place (with jit.hip's directory helpers it calls) where the planning / per-block switches are read: setup calls only.
EnvSwitches read_env_switches() {
    EnvSwitches s;
    s.xcd_remap = env_int("DSPFX_XCD_REMAP");
    s.mix_tail = env_int("DSPFX_MIX_TAIL");
    s.fast_div = env_int("DSPFX_FAST_DIV");
    s.jit = env_int("DSPFX_JIT");
    s.jit_async = env_int("DSPFX_JIT_ASYNC");
    s.ts_tail = env_int("DSPFX_TS_TAIL");
    s.menu_ring_reserve = env_int("DSPFX_MENU_RING_RESERVE");
    if (const char *v = getenv("DSPFX_VARIANT")) {
        s.has_variant = true;
        const char *q;
        if ((q = strstr(v, "f="))) s.pref.f = atoi(q + 2);
        if ((q = strstr(v, "cpl="))) s.pref.cpl = atoi(q + 4);
        if ((q = strstr(v, "static="))) s.pref.stat = atoi(q + 7);
        if ((q = strstr(v, "ts="))) s.variant_ts = atoi(q + 3);
        s.variant_static0 = strstr(v, "static=0") != nullptr;
    }
    s.headers_dir = jit_headers_dir();
    s.cache_dir = jit_cache_dir();
    return s;
}

bool stage_fast_div(const dspfx_engine *e, const Stage &st, bool have_device) {
    const bool off = e->env.fast_div == 0;
    if (!divisor_is_fast(e->hop_div, have_device, off)) return false;
    for (int i = 0; i < st.count; ++i)
        if (!node_divisors_fast(e->nodes[st.first + i], have_device, off)) return false;
    return true;
}

bool node_needs_libm(const Node &n) {
    if (n.d.kind == DSPFX_OVERDRIVE || n.d.kind == DSPFX_CHEBYSHEV || n.d.kind == DSPFX_SIGNAL_GEN) return true;
    return n.d.kind == DSPFX_DISTORT &&
           (n.d.mode == DSPFX_DIST_TANH || n.d.mode == DSPFX_DIST_SIN || n.d.mode == DSPFX_DIST_ATAN);
}

// Does the library hold a compiled-in specialisation of this stage's shape?
static bool has_static_variant(const dspfx_engine *e, const Stage &st, const std::vector<const Variant *> &all) {
    int sigs[MAX_SLOTS];
    stage_sigs(e, st, sigs);
    for (const Variant *v : all) {
        if (v->guard || v->mod || v->ts || v->sigs[0] == SIG_DYN || v->n_slots != st.count) continue;
        bool ok = true;
        for (int i = 0; i < st.count && ok; ++i) {
            const Node &n = e->nodes[st.first + i];
            ok = v->sigs[i] == sig(n.d.kind, n.d.kind == DSPFX_DISTORT ? n.d.mode : 0, node_hop(e, st.first + i));
        }
        if (ok) return true;
    }
    return false;
}

// *pending: no specialised kernel yet, the background compiler is to make one (request_async_jit): the interpreter
// instantiation returned then has the channels per lane OF THAT KERNEL, so that the bus' rows of partial sums -- one per
// workgroup of 256 x cpl channels, or per wave below TS_MAX_CHANNELS -- are the same before and after the switch.
const Variant *pick_variant(const dspfx_engine *e, const Stage &st, bool *pending) {
    std::vector<const Variant *> all;
    collect_variants(all);
    const Pref pref = read_pref(e);
    const uint32_t N = e->desc.channels;
    const Variant *best = nullptr;
    int best_score = -1;
    bool pend = false;
    if (pending) *pending = false;
    // no compiled-in specialisation: one made at run time -- already in this process or in the disk cache, compiled now
    // (JP_SYNC), or on its way (JP_ASYNC)
    const JitPolicy policy = jit_policy(e);
    const bool jit_ok = policy != JP_OFF && pref.stat != 0 && st.fast_div && st.count >= 1 && st.count <= MAX_SLOTS && !e->graph_mode &&
                        !(st.fast_div && pref.stat != 0 && has_static_variant(e, st, all));
    if (jit_ok) {
        if (const Variant *j = jit_variant(e, st, false, policy == JP_SYNC ? JIT_COMPILE : JIT_DISK)) return j;
        pend = policy == JP_ASYNC && !st.jit_failed && N >= 64u * (unsigned)jit_std_cpl(e, st.count);
    }
    if (pending) *pending = pend;
    for (const Variant *v : all) {
        if (v->guard || v->mod || v->ts) continue;
        const bool is_dyn = v->sigs[0] == SIG_DYN;
        if (is_dyn) {
            bool need = false;
            for (int i = 0; i < st.count; ++i) need = need || node_needs_libm(e->nodes[st.first + i]);
            if (v->libm != need) continue;
        }
        if (!is_dyn) {
            if (pref.stat == 0) continue;
            if (!st.fast_div) continue;          // static kernels are built with the fast division only
            if (v->n_slots != st.count) continue;
            bool ok = true;
            for (int i = 0; i < st.count && ok; ++i) {
                const Node &n = e->nodes[st.first + i];
                const int mode = n.d.kind == DSPFX_DISTORT ? n.d.mode : 0;
                ok = v->sigs[i] == sig(n.d.kind, mode, node_hop(e, st.first + i));
            }
            if (!ok) continue;
        }
        if (!is_dyn || v->cpl > 1) {     // the one-channel interpreter is the fallback for every N (the tail launch covers N < 64)
            if (N % (64u * v->cpl) != 0 && N < 64u * v->cpl) continue;
            if (N % v->cpl != 0) continue;   // vector loads need aligned rows
        }
        int score = is_dyn ? 0 : 100;
        // defaults chosen from measurements on MI355X (profiles/): see DESIGN.md
        // Few channels (<= 2 waves per SIMD at one channel per lane): nothing hides a wave's memory latency, so
        // spread over all SIMDs (CPL 1) and keep more loads in flight per wave (F 16): profiles/r01_small_n.txt
        // Round 3 sweep of every compiled (F, CPL) at 114688 ... 262144 channels with tuned placement
        // (tools/r03_midn_variants.py, profiles/r03_small_n.txt): F 16 only BELOW 131072 (5-node chain at 131072: 60.4 us at
        // F 16, 53.6 at F 8; round 4: only below 65536, jit_std_f), and one channel per lane up to ~229000 (163840 channels: 74.0 us at CPL 2, 63.3 at CPL 1;
        // 196608: 79.8 / 75.7; 262144: 92.5 / 95.2 -- from there two channels per lane win).
        const bool few = N <= 131072u;
        const int want_f = is_dyn ? 8 : jit_std_f(e, false, st.count);
        // interpreter: two channels per lane halve the per-chunk interpretive overhead per sample (0.4275 -> 0.383 ms on
        // the 5-node chain, 0.487 -> 0.415 on an 8-node one, 0.612 -> 0.490 with a Tanh node)
        const int want_cpl = is_dyn ? (pend ? jit_std_cpl(e, st.count) : (few ? 1 : 2)) : jit_std_cpl(e, st.count);
        if (pref.f > 0 ? v->f == pref.f : v->f == want_f) score += 10;   // A/B: profiles/r01_ab_dyn.txt
        if (pref.cpl > 0 ? v->cpl == pref.cpl : v->cpl == want_cpl) score += 5;
        if (score > best_score) {
            best_score = score;
            best = v;
        }
    }
    return best;
}

// Few channels (at most two waves per SIMD at one channel per lane): the time-sliced kernel of the same chain shape, if the
// library has one.  DSPFX_VARIANT="ts=0" switches it off, "ts=1" forces it at any size (A/B runs).
const Variant *pick_ts_variant(const dspfx_engine *e, const Stage &st, bool *pending) {
    const uint32_t N = e->desc.channels;
    if (pending) *pending = false;
    const int want = e->env.variant_ts;
    // Measured crossover against the standard kernels (tools/r03_ts_threshold.py, placement tuned, three engines each;
    // profiles/r03_small_n.txt): 98304 channels for chains of up to three nodes (38.5 against 44.2 us there, a tie at 114688),
    // 65536 for longer ones (more registers per wave, fewer co-resident workgroups: 32.8 against 33.7 us, a tie at 73728).
    // (round 4: where the standard kernel takes two channels per lane -- tiled engines of a whole number of 128-channel waves,
    // short chains -- it is one workgroup per CU from 65536 channels on and beats the time-sliced kernel's second round: jit_std_cpl)
    const uint32_t ts_max = (st.count <= 3 && !(N > 65536u && jit_std_cpl(e, st.count) == 2)) ? TS_MAX_CHANNELS : 65536u;
    if (want == 0 || (want < 0 && N > ts_max) || !st.fast_div || st.count < 1 || st.count > MAX_SLOTS) return nullptr;
    const Pref pref = read_pref(e);
    std::vector<const Variant *> all;
    collect_variants(all);
    const Variant *best = nullptr;
    for (const Variant *v : all) {
        if (!v->ts || v->guard || v->n_slots != st.count || N < 64u * (unsigned)v->cpl || N % (unsigned)v->cpl) continue;
        bool ok = true;
        for (int i = 0; i < st.count && ok; ++i) {
            const Node &n = e->nodes[st.first + i];
            const bool has_mode = n.d.kind == DSPFX_DISTORT || n.d.kind == DSPFX_SIGNAL_GEN;
            ok = v->sigs[i] == sig(n.d.kind, has_mode ? n.d.mode : 0, node_hop(e, st.first + i));
        }
        if (!ok) continue;
        // two channels per lane (half the load / store instructions per byte) once there are still two workgroups per CU
        // at that width and the chain is short enough for the doubled registers: 3-node chain at 65536 channels 29.0 -> 27.6 us
        // (one channel per lane holds one resident round up to 49152 channels -- three workgroups per CU; from there two per
        // lane: 53248 channels 27.8 us at one, 21.1 at two; 49152 itself, since the rows go through buffer descriptors: 20.5-21.1
        // at one, 19.2 at two; 32768: 13.8 / 16.8 -- profiles/r04_midn.txt, section G)
        const int want_cpl = pref.cpl > 0 ? pref.cpl : (st.count <= 3 && N >= 49152u ? 2 : 1);
        if (!best || (v->cpl == want_cpl && best->cpl != want_cpl)) best = v;
    }
    if (best) return best;
    // no compiled-in time-sliced kernel for this chain shape: one made at run time, like the standard kernel (one channel per lane)
    const JitPolicy policy = jit_policy(e);
    if (policy == JP_OFF || pref.stat == 0 || N < 64u || e->graph_mode) return nullptr;
    int sigs[MAX_SLOTS];
    stage_sigs(e, st, sigs);
    const JitKernel *k = jit_get(e->device, sigs, st.count, 32, 1, false, true, false, policy == JP_SYNC ? JIT_COMPILE : JIT_DISK);
    if (!k && pending) *pending = policy == JP_ASYNC && !st.jit_failed;
    return k ? &k->var : nullptr;
}

// The channels a whole-wave launch leaves over (N % (64 cpl) of them) used to go through ONE wave per 64 channels of the
// guarded interpreter, walking the block chunk by chunk on an otherwise idle chip: 43 us behind every block, at any N.
// For blocks of exactly 128 frames the guarded time-sliced kernel of the same chain shape takes them instead -- four slices in
// parallel, every load issued at once -- when the library has one or the run-time compiler is in use for this engine.
// DSPFX_TS_TAIL=0 keeps the interpreter (A/B runs, tests: bit-identical).
const Variant *pick_ts_tail_variant(const dspfx_engine *e, const Stage &st, bool *pending) {
    const uint32_t N = e->desc.channels;
    if (pending) *pending = false;
    if (e->env.ts_tail == 0 || !st.fast_div || st.count < 1 || st.count > MAX_SLOTS) return nullptr;
    unsigned cpl = st.var ? (unsigned)st.var->cpl : 1u;
    if (st.var_ts) cpl = std::max(cpl, (unsigned)st.var_ts->cpl);
    if (N % (64u * cpl) == 0) return nullptr;          // no launch of this engine leaves channels over
    int sigs[MAX_SLOTS];
    stage_sigs(e, st, sigs);
    std::vector<const Variant *> all;
    collect_variants(all);
    for (const Variant *v : all) {
        if (!v->ts || !v->guard || v->n_slots != st.count) continue;
        bool ok = true;
        for (int i = 0; i < MAX_SLOTS && ok; ++i) ok = v->sigs[i] == sigs[i];
        if (ok) return v;
    }
    const JitPolicy policy = jit_policy(e);
    const Pref pref = read_pref(e);
    if (policy == JP_OFF || pref.stat == 0 || e->graph_mode) return nullptr;
    const JitKernel *k = jit_get(e->device, sigs, st.count, 32, 1, false, true, true, policy == JP_SYNC ? JIT_COMPILE : JIT_DISK);
    if (!k && pending) *pending = policy == JP_ASYNC && !st.jit_failed;
    return k ? &k->var : nullptr;
}

// Kernels of this stage's shape that neither this process nor the disk cache holds: have the background thread compile them
// (jit.hip) and adopt them when they are ready.  The engine serves its blocks meanwhile -- on the interpreter, or on whichever
// of the kernels it already has.
void request_async_jit(const dspfx_engine *e, const Stage &st, bool want_std, bool want_ts, bool want_tail) {
    if (!(want_std || want_ts || want_tail) || jit_policy(e) != JP_ASYNC) return;
    auto job = std::make_shared<AsyncJit>();
    job->device = e->device;
    job->n_slots = st.count;
    stage_sigs(e, st, job->sigs);
    job->want_std = want_std;
    job->f_std = jit_std_f(e, false, st.count);
    job->cpl_std = jit_std_cpl(e, st.count);
    job->want_ts = want_ts;
    job->want_tail = want_tail;
    st.async = job;
    async_jit_submit(job);
}

// The stage's kernel with control ports, the first time a port is connected (run_subblock): from the caches, compiled now
// (JP_SYNC), or by the background thread -- the control-port interpreter serves meanwhile.
void request_mod_kernel(dspfx_engine *e, const Stage &st) {
    st.var_mod_tried = true;
    const JitPolicy policy = jit_policy(e);
    if (e->env.variant_static0 || policy == JP_OFF || e->graph_mode) return;
    JitDirScope dirs(e);                 // (run_subblock calls this on a stage's first connected control port: no getenv from here)
    st.var_mod = jit_variant(e, st, true, policy == JP_SYNC ? JIT_COMPILE : JIT_DISK);
    if (st.var_mod || policy != JP_ASYNC || st.count < 1 || st.count > MAX_SLOTS || !st.fast_div || e->desc.channels < 64u * (unsigned)jit_std_cpl(e, st.count)) return;
    auto job = std::make_shared<AsyncJit>();
    job->device = e->device;
    job->n_slots = st.count;
    stage_sigs(e, st, job->sigs);
    job->want_mod = true;
    job->f_mod = jit_std_f(e, true, st.count);
    job->cpl_std = jit_std_cpl(e, st.count);
    st.mod_two = job->cpl_std == 2 ? 1 : 0;      // the control-port interpreter runs with THAT kernel's channels per lane, whether it ever arrives or not
    st.async_mod = job;
    async_jit_submit(job);
}

const Variant *pick_variant(const dspfx_engine *e, const Stage &st, bool *pending);
// run_subblock, at a block boundary: the background compiler is done with this stage's shape
void adopt_async_jit(dspfx_engine *e, const Stage &st) {
    if (st.async_mod && st.async_mod->ready.load(std::memory_order_acquire) != 0) {
        const std::shared_ptr<AsyncJit> job = st.async_mod;
        st.async_mod.reset();
        if (job->k_mod) st.var_mod = &job->k_mod->var;
        else e->jit_unavailable = true;
    }
    if (!st.async || st.async->ready.load(std::memory_order_acquire) == 0) return;
    const std::shared_ptr<AsyncJit> job = st.async;
    st.async.reset();
    if (job->ready.load(std::memory_order_acquire) < 0 || (job->want_std && !job->k_std)) {
        // no run-time compiler here: the interpreter stays -- the SAME instantiation the stage started on (the coming kernel's
        // channels per lane), for the life of this plan: another one would cut the bus' rows differently and change its f32
        // summation order at a block nobody can predict; the next dspfx_chain_set picks the best one for the size
        e->jit_unavailable = true;
        st.jit_failed = true;
    }
    if (job->k_std) st.var = &job->k_std->var;
    if (job->k_ts) st.var_ts = &job->k_ts->var;
    if (job->k_tail) st.var_ts_tail = &job->k_tail->var;
}

// Chain engines: may a fusable run of more than MAX_SLOTS nodes become one generated kernel?  The conditions of the
// run-time specialised chain kernels (jit_variant), whole waves only, and not after control ports were used (those are
// evaluated by the chain kernels).
bool long_stage_wanted(const dspfx_engine *e) {
    if (e->no_long) return false;
    const uint32_t N = e->desc.channels;
    const int jit_mode = e->env.jit;
    if (!(jit_mode == 1 || (jit_mode != 0 && N > 131072u))) return false;
    if (read_pref(e).stat == 0) return false;
    return N % 64u == 0;
}

int plan(dspfx_engine *e) {
    HIPCHK(e, hipSetDevice(e->device));   // divisor checks run there, run-time compiled modules are loaded there
    JitDirScope dirs(e);                  // a re-plan at a block boundary (mode store) reads the engine's snapshot, not the environment
    for (const Stage &st : e->stages) {
        if (st.async) st.async->abandoned.store(true, std::memory_order_release);
        if (st.async_mod) st.async_mod->abandoned.store(true, std::memory_order_release);
    }
    e->stages.clear();
    e->jit_unavailable = false;
    e->has_fuzz = false;
    e->has_siggen = false;
    e->min_delay = 0xffffffffu;
    const int n = (int)e->nodes.size();
    int i = 0;
    while (i < n) {
        Stage st{};
        if (fusable(e->nodes[i])) {
            st.type = ST_FUSED;
            st.first = i;
            int limit = e->graph_mode ? GRAPH_SLOTS : MAX_SLOTS;
            if (!e->graph_mode && long_stage_wanted(e)) {
                // A fusable run longer than one chain launch holds: up to GRAPH_SLOTS of its nodes become one generated
                // kernel (the chain as a graph) instead of two chain launches with a round trip through memory in
                // between (12 nodes: 0.575 -> 0.429 ms).  Add / Mix read the engine's side input from memory, which
                // that kernel does not do: a run with one of them is cut as before.
                int j = i;
                bool mixers = false;
                while (j < n && fusable(e->nodes[j]) && j - i < GRAPH_SLOTS) {
                    mixers = mixers || e->nodes[j].d.kind == DSPFX_ADD || e->nodes[j].d.kind == DSPFX_MIX;
                    ++j;
                }
                if (j - i > MAX_SLOTS && !mixers) limit = GRAPH_SLOTS;
            }
            while (i < n && fusable(e->nodes[i]) && i - st.first < limit) ++i;
            st.count = i - st.first;
        } else {
            st.type = e->nodes[i].d.kind == DSPFX_FIR ? ST_FIR : ST_FUZZ;
            st.first = i;
            st.count = 1;
            ++i;
        }
        e->stages.push_back(st);
    }
    // the mix bus is reduced in the epilogue of a fused stage: make sure one is last
    if (e->stages.empty() || e->stages.back().type != ST_FUSED) {
        Stage st{};
        st.type = ST_FUSED;
        st.first = n;
        st.count = 0;
        e->stages.push_back(st);
    }
    if (e->graph_mode && (e->stages.size() != 1 || e->stages[0].type != ST_FUSED || e->stages[0].count != n))
        return fail(e, DSPFX_ERR_UNSUPPORTED, "graph has a node that cannot be fused (FIR, Fuzz) or more than %d nodes", GRAPH_SLOTS);
    for (Stage &st : e->stages)
        if (st.type == ST_FUSED) {
            st.fast_div = stage_fast_div(e, st);
            bool pend_std = false, pend_ts = false, pend_tail = false;
            st.var = (e->graph_mode || st.count > MAX_SLOTS) ? graph_variant(e, st) : pick_variant(e, st, &pend_std);
            if (!st.var && !e->graph_mode && st.count > MAX_SLOTS) {   // no run-time compiler: cut the run as usual
                e->no_long = true;
                return plan(e);
            }
            if (!st.var)
                return fail(e, DSPFX_ERR_UNSUPPORTED, e->graph_mode ? "the graph kernel could not be compiled (hiprtc / csrc headers unavailable)"
                                                                    : "no kernel variant for stage");
            st.var_ts = (e->graph_mode || st.count > MAX_SLOTS) ? nullptr : pick_ts_variant(e, st, &pend_ts);
            st.var_ts_tail = (e->graph_mode || st.count > MAX_SLOTS) ? nullptr : pick_ts_tail_variant(e, st, &pend_tail);
            request_async_jit(e, st, pend_std, pend_ts, pend_tail);
        }
    for (const Node &nd : e->nodes) {
        if (nd.d.kind == DSPFX_DISTORT && nd.d.mode == DSPFX_DIST_FUZZ) e->has_fuzz = true;
        if (nd.d.kind == DSPFX_REVERB) e->min_delay = std::min(e->min_delay, nd.D);
        if (nd.d.kind == DSPFX_SIGNAL_GEN) e->has_siggen = true;
    }
    return DSPFX_OK;
}

}  // namespace dspfx_host
