// placement.hip -- where the delay rings' 128-row groups land in HBM decides how fast they stream next to the caller's
// sample buffers (placement): the setup-time probe of a new ring and dspfx_tune_placement.  See engine.h.
#include "engine.h"

using namespace dspfx;
using namespace dspfx_host;

namespace dspfx_host {

// Placement tuning of a delay ring (see chain_kernels.hip.h, ring layout).  Some physical HBM regions
// stream ~18 % slower under the chain kernel's access pattern (every resident workgroup walking its own
// 128 KiB tile); the effect is stable over time and independent of the in/out buffers, but a plain
// streaming sweep does not show it (profiles/r01_placement.txt), so the probe IS the delay node's kernel:
// each candidate group is timed as a one-node REVERB launch over engine-owned scratch in/out.  As many
// extra candidates as memory allows (at most as many as the ring has groups) are allocated, all are
// timed, the fastest are kept.  Setup-time only; DSPFX_RING_TUNE=0/1 forces it off/on (default: rings
// whose groups are >= 64 MiB).
// Large streamed buffers (delay-ring groups, engine-owned sample buffers).  DSPFX_CONTIG=1 asks the driver for
// physically contiguous VRAM, which lets the page tables use their largest fragment size (TLB reach).
hipError_t big_alloc(void **p, size_t bytes) {
    static const int contig = [] { const char *c = getenv("DSPFX_CONTIG"); return c ? atoi(c) : 0; }();
    if (contig && bytes >= ((size_t)2 << 20)) {
        if (hipExtMallocWithFlags(p, bytes, hipDeviceMallocContiguous) == hipSuccess) return hipSuccess;
        (void)hipGetLastError();
    }
    return hipMalloc(p, bytes);
}

int tune_ring(dspfx_engine *e, Node &n) {
    const char *tv = getenv("DSPFX_RING_TUNE");
    const int mode = tv ? atoi(tv) : -1;
    const size_t gbytes = n.group_floats * sizeof(float);
    if (mode == 0 || (mode < 0 && gbytes < ((size_t)64 << 20))) return DSPFX_OK;
    const uint32_t N = e->desc.channels;
    if (N < 64 || !e->dyn) return DSPFX_OK;
    const size_t G = n.groups.size();
    size_t free_b = 0, total_b = 0;
    HIPCHK(e, hipMemGetInfo(&free_b, &total_b));
    const size_t scratch_bytes = 2 * gbytes;                     // in + out blocks of 128 frames
    const size_t reserve = (size_t)8 << 30;                      // leave room for the caller's buffers
    (void)total_b;
    float *scratch = nullptr, **d_one = nullptr;
    hipEvent_t a = nullptr, b = nullptr;
    if (hipMalloc((void **)&scratch, scratch_bytes) != hipSuccess) return DSPFX_OK;   // no room: skip tuning
    HIPCHK(e, hipMemset(scratch, 0, scratch_bytes));
    HIPCHK(e, hipMalloc((void **)&d_one, sizeof(float *)));
    HIPCHK(e, hipEventCreate(&a));
    HIPCHK(e, hipEventCreate(&b));
    ChainArgs ca;
    memset(&ca, 0, sizeof ca);
    ca.in = scratch;
    ca.out = scratch + n.group_floats;
    ca.N = N;
    ca.nframes = RING_GROUP_ROWS;
    if (e->desc.tile_channels) {
        const uint32_t W = e->desc.tile_channels;
        ca.w_shift = (unsigned)__builtin_ctz(W);
        ca.w_mask = W - 1;
        ca.ld = W;
        ca.io_tile_stride = (size_t)RING_GROUP_ROWS * W;
    } else {
        ca.w_shift = 31;
        ca.w_mask = 0x7fffffffu;
        ca.ld = N;
    }
    ca.hop_div = e->hop_div;
    ca.hop_rc = 1.0 / (double)e->hop_div;
    ca.third_rc = 1.0 / 3.0;
    ca.fast_div = 1;
    ca.xcd_remap = 1;
    ca.n_slots = 1;
    ca.slot[0].kind = DSPFX_REVERB;
    ca.slot[0].p[0] = 0.5f;
    ca.slot[0].groups = d_one;
    ca.slot[0].D = RING_GROUP_ROWS;
    const uint32_t n_main = N - N % 64u;
    ca.n_launch = n_main;
    const unsigned grid = (n_main + WG - 1) / WG;
    auto probe = [&](float *gptr, float &best) -> int {
        HIPCHK(e, hipMemcpy(d_one, &gptr, sizeof(float *), hipMemcpyHostToDevice));
        best = 1e30f;
        for (int rep = 0; rep < 3; ++rep) {   // rep 0 warms TLB/clocks
            (void)hipEventRecord(a, nullptr);
            (void)launch_variant(e->dyn, ca, grid, WG, 0, nullptr);
            (void)hipEventRecord(b, nullptr);
            HIPCHK(e, hipEventSynchronize(b));
            float ms = 0.0f;
            (void)hipEventElapsedTime(&ms, a, b);
            if (rep) best = std::min(best, ms);
        }
        return DSPFX_OK;
    };
    // The ring's own groups first; candidates are allocated only for groups in the slow placement mode (SLOW above the
    // fastest), about 1.8 per slow group and round -- a fresh device, where every group is fast, allocates eight scouts.
    const float SLOW = 1.06f;
    std::vector<float> t(G, 0.0f);
    int rc = DSPFX_OK;
    for (size_t g = 0; g < G && rc == DSPFX_OK; ++g) rc = probe(n.groups[g], t[g]);
    std::vector<std::pair<float, float *>> pool;
    auto more_candidates = [&](size_t want) {
        size_t fb = 0, tb = 0;
        if (hipMemGetInfo(&fb, &tb) != hipSuccess) return;
        const size_t can = fb > reserve ? (fb - reserve) / gbytes : 0;
        for (size_t k = 0; k < std::min(want, can) && rc == DSPFX_OK; ++k) {
            float *g = nullptr;
            if (big_alloc((void **)&g, gbytes) != hipSuccess) { (void)hipGetLastError(); break; }
            float ms = 0.0f;
            rc = probe(g, ms);
            pool.emplace_back(ms, g);
        }
    };
    if (rc == DSPFX_OK) more_candidates(std::min<size_t>(8, G));
    int replaced = 0;
    size_t n_alloc = pool.size();
    if (rc == DSPFX_OK) {
        float t_ref = *std::min_element(t.begin(), t.end());
        for (auto &pr : pool) t_ref = std::min(t_ref, pr.first);
        for (int round = 0; round < 4 && rc == DSPFX_OK; ++round) {
            std::vector<size_t> slow;
            for (size_t g = 0; g < G; ++g)
                if (t[g] > SLOW * t_ref) slow.push_back(g);
            std::sort(slow.begin(), slow.end(), [&](size_t x, size_t y) { return t[x] > t[y]; });
            std::sort(pool.begin(), pool.end());
            size_t used = 0;
            for (size_t g : slow) {
                if (used >= pool.size() || pool[used].first > SLOW * t_ref) break;
                std::swap(n.groups[g], pool[used].second);
                t[g] = pool[used].first;
                pool[used].first = 1e30f;
                ++used;
                ++replaced;
            }
            size_t still = 0;
            for (size_t g = 0; g < G; ++g) still += t[g] > SLOW * t_ref ? 1 : 0;
            if (!still) break;
            const size_t before = pool.size();
            more_candidates(still + (still * 4 + 4) / 5);
            n_alloc += pool.size() - before;
            if (pool.size() == before) break;
            for (size_t i = before; i < pool.size(); ++i) t_ref = std::min(t_ref, pool[i].first);
        }
    }
    if (getenv("DSPFX_RING_TUNE_DEBUG")) {
        fprintf(stderr, "ring probe: %zu groups, %zu candidates allocated, %d re-placed; ms now:", G, n_alloc, replaced);
        for (float v : t) fprintf(stderr, " %.3f", v);
        fprintf(stderr, "\n");
    }
    for (auto &pr : pool) (void)hipFree(pr.second);
    n.ring_replaced = replaced;
    (void)hipFree(scratch);
    (void)hipFree(d_one);
    (void)hipEventDestroy(a);
    (void)hipEventDestroy(b);
    if (rc) return rc;
    n.zero_left = n.D;                 // probing wrote into the groups: the ring is still the zero ring it was (Node::zero_left)
    HIPCHK(e, hipMemcpy(n.d_groups, n.groups.data(), n.groups.size() * sizeof(float *), hipMemcpyHostToDevice));
    return DSPFX_OK;
}

}  // namespace dspfx_host

// Placement tuning against the caller's own buffers (see include/dspfx.h).  Every candidate 128-row group of every
// large delay ring is timed with the REAL chain (all stages, the engine's chosen kernels) reading `in` and writing
// `out`: the node temporarily becomes a 128-row ring made of that one group.  The fastest groups are kept.
// The engine's DSP state survives.  The probes run whole blocks through EVERY node, so everything a block writes is
// parked first and put back afterwards:
//   * filter / generator / envelope rows: snapshotted, restored at the end;
//   * every delay ring (the probed one, the small ones, the ones probed earlier or later): its position is put back
//     before every probe run, so all probes overwrite the same n_frames rows -- parked once, restored at the end;
//   * the ring being probed: each of its groups is parked in a scratch group while it stands in as the one-group ring,
//     and put back (or moved into the candidate that replaces it, at the same ring position);
//   * FIR nodes: the rows the probes' samples land in, the non-finite flags, the fill-phase sums and the host-side
//     deque model (fir_park / fir_rewind / fir_unpark).
namespace dspfx_host {
struct TuneGuard {   // whatever happens inside the probe loop, every node gets its real geometry / position back and nothing leaks
    dspfx_engine *e;
    Node *n = nullptr;
    uint32_t D0 = 0, min0 = 0;
    float **table0 = nullptr, **d_one = nullptr;
    float *park = nullptr;
    hipEvent_t ea = nullptr, eb = nullptr;
    std::vector<float *> extras;            // candidates allocated here and not (yet) adopted by the ring
    std::vector<std::pair<float *, size_t>> snaps;   // device copies of node state: (copy, node index)
    std::vector<uint32_t> pos0;             // ring position of every node on entry
    std::vector<uint32_t> zero0;            // ... and its count of frames still reading a cleared ring (Node::zero_left)
    std::vector<float *> rows;              // per node: the parked rows [n_frames][N] of its delay ring (or null)
    std::vector<FirPark> firs;              // per node
    explicit TuneGuard(dspfx_engine *e_) : e(e_), pos0(e_->nodes.size(), 0), zero0(e_->nodes.size(), 0), rows(e_->nodes.size(), nullptr), firs(e_->nodes.size()) {
        for (size_t i = 0; i < e->nodes.size(); ++i) {
            pos0[i] = e->nodes[i].pos;
            zero0[i] = e->nodes[i].zero_left;
        }
    }
    void arm(Node &node) {
        n = &node;
        D0 = node.D;
        min0 = e->min_delay;
        table0 = node.d_groups;
    }
    void disarm() {
        if (!n) return;
        n->D = D0;
        n->d_groups = table0;
        n->probe_group = nullptr;
        e->min_delay = min0;
        n = nullptr;
    }
    // host-side positions as on entry (the node being probed: a one-group ring at position 0)
    void rewind() {
        for (size_t i = 0; i < e->nodes.size(); ++i) {
            Node &m = e->nodes[i];
            if (m.d.kind == DSPFX_REVERB) {
                m.pos = &m == n ? 0 : pos0[i];
                m.zero_left = zero0[i];       // the probes' blocks count it down like any block
            }
            if (m.d.kind == DSPFX_FIR && firs[i].rows) fir_rewind(m.fir, firs[i]);
        }
    }
    ~TuneGuard() {
        disarm();
        rewind();
        for (float *g : extras)
            if (g) (void)hipFree(g);
        for (auto &sn : snaps)
            if (sn.first) (void)hipFree(sn.first);
        for (float *r : rows)
            if (r) (void)hipFree(r);
        for (FirPark &f : firs) fir_park_free(f);
        if (d_one) (void)hipFree(d_one);
        if (park) (void)hipFree(park);
        if (ea) (void)hipEventDestroy(ea);
        if (eb) (void)hipEventDestroy(eb);
        (void)hipGetLastError();
    }
};
}  // namespace dspfx_host

extern "C" int dspfx_tune_placement(dspfx_engine *e, const float *in, const float *side, float *out, uint32_t n_frames,
                                    void *stream) {
    if (!e) return DSPFX_ERR_INVALID;
    hipStream_t s = (hipStream_t)stream;
    ApiScope api(e, true, s);
    if (api.rc) return api.rc;
    if (!in || !out) return fail(e, DSPFX_ERR_INVALID, "in/out must not be null");
    if (n_frames == 0 || n_frames > e->desc.max_frames || n_frames > RING_GROUP_ROWS)
        return fail(e, DSPFX_ERR_INVALID, "n_frames must be 1..%u here", std::min<uint32_t>(e->desc.max_frames, RING_GROUP_ROWS));
    if (e->collect_due || e->mp_count) return fail(e, DSPFX_ERR_STATE, "flush the mix pipeline before tuning");
    bool any = false;
    for (const Node &n : e->nodes) any = any || (n.d.kind == DSPFX_REVERB && n.group_floats * sizeof(float) >= ((size_t)64 << 20));
    if (!any) return DSPFX_OK;               // nothing large enough to be placement-sensitive: state untouched
    if (e->graph_mode)                       // a region kernel with extra blocks: its buffers are not all known here
        for (const dspfx_graph_link &l : e->wiring)
            if (graph_input_block(l.src) >= 2 || l.dst > (int)e->nodes.size()) return DSPFX_OK;
    const uint32_t N = e->desc.channels;
    const uint32_t W = e->desc.tile_channels ? e->desc.tile_channels : N;
    TuneGuard tg(e);
    if (hipEventCreate(&tg.ea) != hipSuccess || hipEventCreate(&tg.eb) != hipSuccess) return fail(e, DSPFX_ERR_HIP, "hipEventCreate failed");
    HIPCHK(e, hipStreamSynchronize(s));
    // park what a block writes: small per-channel state rows, the block's rows of every delay ring, FIR histories
    for (size_t i = 0; i < e->nodes.size(); ++i) {
        Node &n = e->nodes[i];
        if (n.d.kind == DSPFX_REVERB) {
            if (hipMalloc((void **)&tg.rows[i], (size_t)n_frames * N * sizeof(float)) != hipSuccess) return fail(e, DSPFX_ERR_OOM, "no room to park ring rows");
            launch_ring_copy(n.d_groups, tg.rows[i], N, W, n.D, n.pos, n_frames, true, s);
            HIPCHK(e, hipGetLastError());
        } else if (n.d.kind == DSPFX_FIR) {
            const int rc = fir_park(n.fir, n_frames, s, tg.firs[i]);
            if (rc) return fail(e, rc, "FIR: %s", fir_last_error());
        } else if (n.state && n.state_bytes) {
            float *copy = nullptr;
            if (hipMalloc((void **)&copy, n.state_bytes) != hipSuccess) return fail(e, DSPFX_ERR_OOM, "no room for a state snapshot");
            tg.snaps.emplace_back(copy, i);
            if (hipMemcpyAsync(copy, n.state, n.state_bytes, hipMemcpyDeviceToDevice, s) != hipSuccess) return fail(e, DSPFX_ERR_HIP, "state snapshot failed");
        }
    }
    int rc = DSPFX_OK;
    const size_t tile_frames = e->desc.max_frames;   // the buffers are laid out like a full block of the engine
    const auto tick = [] { return std::chrono::steady_clock::now(); };
    const auto ms_since = [](std::chrono::steady_clock::time_point t0) { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count(); };
    const bool debug = getenv("DSPFX_RING_TUNE_DEBUG") != nullptr;
    for (Node &n : e->nodes) {
        if (n.d.kind != DSPFX_REVERB) continue;
        const size_t gbytes = n.group_floats * sizeof(float);
        if (gbytes < ((size_t)64 << 20)) continue;
        const size_t G = n.groups.size();
        if (!tg.park && big_alloc((void **)&tg.park, gbytes) != hipSuccess) {
            (void)hipGetLastError();
            return fail(e, DSPFX_ERR_OOM, "no room to park a ring group (%zu MiB)", gbytes >> 20);
        }
        if (!tg.d_one && hipMalloc((void **)&tg.d_one, sizeof(float *)) != hipSuccess) return fail(e, DSPFX_ERR_OOM, "hipMalloc failed");
        // the node as a one-group ring (the guard puts the real geometry back on every exit path)
        tg.arm(n);
        n.D = RING_GROUP_ROWS;
        n.d_groups = tg.d_one;
        // one candidate: the real chain, `reps` launches, the best of all but the first (which warms TLB / clocks)
        auto probe = [&](float *gptr, bool live, float &best) -> int {
            if (live && hipMemcpyAsync(tg.park, gptr, gbytes, hipMemcpyDeviceToDevice, s) != hipSuccess) return fail(e, DSPFX_ERR_HIP, "parking a ring group failed");
            if (hipMemcpyAsync(tg.d_one, &gptr, sizeof(float *), hipMemcpyHostToDevice, s) != hipSuccess) return fail(e, DSPFX_ERR_HIP, "hipMemcpyAsync failed");
            if (hipStreamSynchronize(s) != hipSuccess) return fail(e, DSPFX_ERR_HIP, "hipStreamSynchronize failed");   // (&gptr is a stack slot)
            n.probe_group = gptr;
            best = 1e30f;
            for (int rep = 0; rep < 3; ++rep) {
                tg.rewind();
                (void)hipEventRecord(tg.ea, s);
                const int r = run_subblock(e, in, side, out, nullptr, n_frames, (uint32_t)tile_frames, s);
                (void)hipEventRecord(tg.eb, s);
                if (r) return r;
                if (hipEventSynchronize(tg.eb) != hipSuccess) return fail(e, DSPFX_ERR_HIP, "hipEventSynchronize failed");
                float ms = 0.0f;
                (void)hipEventElapsedTime(&ms, tg.ea, tg.eb);
                if (rep) best = std::min(best, ms);
            }
            if (live && hipMemcpyAsync(gptr, tg.park, gbytes, hipMemcpyDeviceToDevice, s) != hipSuccess) return fail(e, DSPFX_ERR_HIP, "restoring a ring group failed");
            return DSPFX_OK;
        };
        // 1. the ring's own groups.  Placement comes in two modes ~18 % apart (placement): when every group is
        //    within SLOW of the fastest one there is nothing to gain and nothing is allocated (a fresh box: 0.25 s).
        const float SLOW = 1.06f;
        const auto t_own = tick();
        std::vector<float> t(G, 0.0f);
        for (size_t g = 0; g < G && rc == DSPFX_OK; ++g) rc = probe(n.groups[g], true, t[g]);
        const double ms_own = ms_since(t_own);
        if (const char *fk = getenv("DSPFX_TUNE_FAKE_SLOW"))            // tests: pretend the first k groups landed in the slow mode
            for (size_t g = 0; g < std::min<size_t>(G, (size_t)atoi(fk)); ++g) t[g] *= 1.25f;
        // 2. a few scouts: is the fastest own group really a fast one (or are ALL of them in the slow mode)?
        auto alloc_candidates = [&](size_t want, std::vector<float *> &got) {
            size_t free_b = 0, total_b = 0;
            if (hipMemGetInfo(&free_b, &total_b) != hipSuccess) return;
            const size_t reserve = ((size_t)8 << 30);            // room for the caller
            const size_t can = free_b > reserve ? (free_b - reserve) / gbytes : 0;
            for (size_t k = 0; k < std::min(want, can); ++k) {
                float *g = nullptr;
                if (big_alloc((void **)&g, gbytes) != hipSuccess) { (void)hipGetLastError(); break; }
                if (hipMemsetAsync(g, 0, gbytes, s) != hipSuccess) { (void)hipFree(g); (void)hipGetLastError(); break; }
                got.push_back(g);
                tg.extras.push_back(g);
            }
        };
        std::vector<std::pair<float, float *>> pool;            // probed candidates that are not part of the ring: (ms, group)
        const auto t_rest = tick();
        size_t n_alloc = 0;
        if (rc == DSPFX_OK) {
            std::vector<float *> scouts;
            alloc_candidates(std::min<size_t>(8, G), scouts);
            n_alloc += scouts.size();
            for (float *g : scouts) {
                float ms = 0.0f;
                rc = probe(g, false, ms);
                if (rc) break;
                pool.emplace_back(ms, g);
            }
        }
        int replaced = 0;
        if (rc == DSPFX_OK) {
            float t_ref = *std::min_element(t.begin(), t.end());
            for (auto &pr : pool) t_ref = std::min(t_ref, pr.first);
            // 3. replace slow groups, slowest first, by fast candidates; allocate more candidates (about 1.8 per group still
            //    slow: roughly 6 in 10 land in the fast mode) until none is slow, memory runs out, or four rounds have passed
            for (int round = 0; round < 4 && rc == DSPFX_OK; ++round) {
                std::vector<size_t> slow;
                for (size_t g = 0; g < G; ++g)
                    if (t[g] > SLOW * t_ref) slow.push_back(g);
                std::sort(slow.begin(), slow.end(), [&](size_t x, size_t y) { return t[x] > t[y]; });
                std::sort(pool.begin(), pool.end());
                size_t used = 0;
                for (size_t g : slow) {
                    if (used >= pool.size() || pool[used].first > SLOW * t_ref) break;
                    float *fresh = pool[used].second;
                    if (hipMemcpyAsync(fresh, n.groups[g], gbytes, hipMemcpyDeviceToDevice, s) != hipSuccess) { rc = fail(e, DSPFX_ERR_HIP, "moving a ring group failed"); break; }
                    std::swap(n.groups[g], pool[used].second);     // the pool now holds the dropped allocation
                    for (float *&x : tg.extras)                     // ... and so does the guard's list of what to free on an error path
                        if (x == fresh) x = pool[used].second;
                    t[g] = pool[used].first;
                    pool[used].first = 1e30f;                      // never picked again
                    ++used;
                    ++replaced;
                }
                if (rc) break;
                size_t still = 0;
                for (size_t g = 0; g < G; ++g) still += t[g] > SLOW * t_ref ? 1 : 0;
                if (!still) break;
                std::vector<float *> more;
                alloc_candidates(still + (still * 4 + 4) / 5, more);
                if (more.empty()) break;
                n_alloc += more.size();
                for (float *g : more) {
                    float ms = 0.0f;
                    rc = probe(g, false, ms);
                    if (rc) break;
                    pool.emplace_back(ms, g);
                    t_ref = std::min(t_ref, ms);
                }
            }
        }
        tg.disarm();
        tg.rewind();
        if (rc) return rc;
        HIPCHK(e, hipStreamSynchronize(s));
        for (auto &pr : pool) (void)hipFree(pr.second);              // dropped originals and unused candidates
        tg.extras.clear();
        n.ring_replaced = replaced;
        HIPCHK(e, hipMemcpyAsync(n.d_groups, n.groups.data(), n.groups.size() * sizeof(float *), hipMemcpyHostToDevice, s));
        HIPCHK(e, hipStreamSynchronize(s));
        if (debug) {
            fprintf(stderr, "placement tuning: %zu ring groups probed in %.0f ms; %zu candidates allocated, %d groups re-placed, %.0f ms; ring groups now (ms):", G, ms_own,
                    n_alloc, replaced, ms_since(t_rest));
            for (float v : t) fprintf(stderr, " %.3f", v);
            fprintf(stderr, "\n");
        }
    }
    // put back what the probes' blocks overwrote
    tg.rewind();
    for (size_t i = 0; i < e->nodes.size(); ++i) {
        Node &n = e->nodes[i];
        if (n.d.kind == DSPFX_REVERB && tg.rows[i]) {
            launch_ring_copy(n.d_groups, tg.rows[i], N, W, n.D, n.pos, n_frames, false, s);
            HIPCHK(e, hipGetLastError());
        } else if (n.d.kind == DSPFX_FIR) {
            const int r = fir_unpark(n.fir, tg.firs[i], s);
            if (r) return fail(e, r, "FIR: %s", fir_last_error());
        }
    }
    for (auto &sn : tg.snaps) {
        Node &n = e->nodes[sn.second];
        if (hipMemcpyAsync(n.state, sn.first, n.state_bytes, hipMemcpyDeviceToDevice, s) != hipSuccess) return fail(e, DSPFX_ERR_HIP, "restoring node state failed");
    }
    // control-port latches are only written while a port is connected: the probes connect none
    HIPCHK(e, hipStreamSynchronize(s));
    return DSPFX_OK;
}
