// state_util.hip -- DSP state export / import, fan-in averaging, and the utilities of the C ABI (noise fill, sync, device checks,
// profiling read-out, algorithmic bytes, dspfx_describe).  See engine.h for the split.
#include "engine.h"

using namespace dspfx;
using namespace dspfx_host;


// -------------------------------------------------------------------- state

extern "C" int dspfx_link_average(dspfx_engine *e, const float *const *srcs, int n_srcs, float *dst, uint32_t n_frames,
                                  void *stream) {
    if (!e) return DSPFX_ERR_INVALID;
    ApiScope api(e);
    if (api.rc) return api.rc;
    if (!dst || n_srcs < 0 || (n_srcs > 0 && !srcs)) return fail(e, DSPFX_ERR_INVALID, "bad link list");
    if (n_srcs > DSPFX_MAX_LINKS) return fail(e, DSPFX_ERR_UNSUPPORTED, "more than %d links into one port", DSPFX_MAX_LINKS);
    if (n_frames == 0) return DSPFX_OK;
    if (n_frames > e->desc.max_frames) return fail(e, DSPFX_ERR_INVALID, "n_frames %u > max_frames %u", n_frames, e->desc.max_frames);
    for (int k = 0; k < n_srcs; ++k)
        if (!srcs[k]) return fail(e, DSPFX_ERR_INVALID, "link %d is null", k);
    HIPCHK(e, hipSetDevice(e->device));
    LinkAvgArgs a;
    memset(&a, 0, sizeof a);
    for (int k = 0; k < n_srcs; ++k) a.src[k] = srcs[k];
    a.n_srcs = n_srcs;
    a.dst = dst;
    a.count = (size_t)n_frames * e->desc.channels;   // element-wise: the same in either layout
    a.div = dspfx_link_divisor((uint64_t)n_srcs);
    launch_link_average(a, (hipStream_t)stream);
    HIPCHK(e, hipGetLastError());
    return DSPFX_OK;
}

extern "C" int64_t dspfx_state_size(const dspfx_engine *e, int node) {
    if (!e) return DSPFX_ERR_INVALID;
    std::lock_guard<std::recursive_mutex> lk(e->api_mu);
    if (node < 0 || node >= (int)e->nodes.size()) return DSPFX_ERR_INVALID;
    const Node &n = e->nodes[(size_t)node];
    if (n.d.kind == DSPFX_FIR) return (int64_t)fir_state_bytes(n.fir);
    if (n.d.kind == DSPFX_REVERB) return (int64_t)n.D * e->desc.channels * (int64_t)sizeof(float);   // canonical [D][N]
    return (int64_t)n.state_bytes;
}

extern "C" int dspfx_state_export(dspfx_engine *e, int node, void *host_dst, size_t size) {
    if (!e || !host_dst) return DSPFX_ERR_INVALID;
    ApiScope api(e);
    if (api.rc) return api.rc;
    if (node < 0 || node >= (int)e->nodes.size()) return fail(e, DSPFX_ERR_INVALID, "node %d out of range", node);
    Node &n = e->nodes[(size_t)node];
    const int64_t need = dspfx_state_size(e, node);
    if ((int64_t)size != need) return fail(e, DSPFX_ERR_INVALID, "state size %zu != %lld", size, (long long)need);
    HIPCHK(e, hipSetDevice(e->device));
    HIPCHK(e, hipDeviceSynchronize());
    if (n.d.kind == DSPFX_FIR) {
        const int rc = fir_state_export(n.fir, host_dst);
        return rc ? fail(e, rc, "FIR: %s", fir_last_error()) : DSPFX_OK;
    }
    if (n.d.kind == DSPFX_REVERB) {   // canonical form: [D][N], row 0 = the oldest sample
        const int rc = ring_rows_copy(e, n, n.pos, n.D, (char *)host_dst, true);
        // rows from before the ring's last clear (Node::zero_left of them, the oldest) ARE zeros as far as any block can tell
        if (rc == DSPFX_OK && n.zero_left) memset(host_dst, 0, (size_t)std::min(n.zero_left, n.D) * e->desc.channels * sizeof(float));
        return rc;
    }
    if (need) HIPCHK(e, hipMemcpy(host_dst, n.state, (size_t)need, hipMemcpyDeviceToHost));
    return DSPFX_OK;
}

extern "C" int dspfx_state_import(dspfx_engine *e, int node, const void *host_src, size_t size) {
    if (!e || !host_src) return DSPFX_ERR_INVALID;
    ApiScope api(e);
    if (api.rc) return api.rc;
    if (node < 0 || node >= (int)e->nodes.size()) return fail(e, DSPFX_ERR_INVALID, "node %d out of range", node);
    Node &n = e->nodes[(size_t)node];
    // a FIR node's blob carries its own length (the deque's, which need not be this engine's current one)
    const int64_t need = n.d.kind == DSPFX_FIR ? fir_state_import_bytes(n.fir, host_src, size) : dspfx_state_size(e, node);
    if ((int64_t)size != need) return fail(e, DSPFX_ERR_INVALID, "state size %zu != %lld", size, (long long)need);
    HIPCHK(e, hipSetDevice(e->device));
    HIPCHK(e, hipDeviceSynchronize());
    if (n.d.kind == DSPFX_FIR) {
        const int rc = fir_state_import(n.fir, host_src);
        return rc ? fail(e, rc, "FIR: %s", fir_last_error()) : DSPFX_OK;
    }
    if (n.d.kind == DSPFX_REVERB) {
        const int rc = ring_rows_copy(e, n, 0, n.D, (char *)const_cast<void *>(host_src), false);
        if (rc) return rc;
    } else if (need) {
        HIPCHK(e, hipMemcpy(n.state, host_src, (size_t)need, hipMemcpyHostToDevice));
    }
    n.pos = 0;
    n.zero_left = 0;
    return settle_null_stream(e);
}

// ---------------------------------------------------------------- utilities

extern "C" int dspfx_fill_noise(dspfx_engine *e, float *dst, uint32_t n_frames, uint32_t n_abs0, uint32_t seed,
                                void *stream) {
    if (!e || !dst) return DSPFX_ERR_INVALID;
    ApiScope api(e);
    if (api.rc) return api.rc;
    Layout lay{};
    if (e->desc.tile_channels) {
        const uint32_t W = e->desc.tile_channels;
        lay = Layout{(unsigned)__builtin_ctz(W), W - 1, W, 0, (size_t)n_frames * W};
    } else {
        lay = Layout{31, 0x7fffffffu, e->desc.channels, 0, 0};
    }
    launch_noise(dst, e->desc.channels, n_frames, (uint32_t)e->desc.channel_offset, n_abs0, seed, lay,
                 (hipStream_t)stream);
    HIPCHK(e, hipGetLastError());
    return DSPFX_OK;
}

extern "C" int dspfx_sync(dspfx_engine *e, void *stream) {
    if (!e) return DSPFX_ERR_INVALID;
    HIPCHK(e, hipSetDevice(e->device));
    HIPCHK(e, hipStreamSynchronize((hipStream_t)stream));
    return DSPFX_OK;
}

extern "C" int dspfx_verify_fast_division(int device, float c, uint64_t *mismatches) {
    if (!mismatches) return DSPFX_ERR_INVALID;
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) return DSPFX_ERR_NO_DEVICE;
    if (device < 0 || device >= ndev || hipSetDevice(device) != hipSuccess) return DSPFX_ERR_INVALID;
    unsigned long long *d = nullptr, h = 0;
    if (hipMalloc((void **)&d, sizeof h) != hipSuccess) return DSPFX_ERR_OOM;
    int rc = DSPFX_ERR_HIP;
    if (hipMemset(d, 0, sizeof h) == hipSuccess && verify_divisor_on_device(c, 1.0 / (double)c, d, nullptr) == 0 &&
        hipMemcpy(&h, d, sizeof h, hipMemcpyDeviceToHost) == hipSuccess) {
        *mismatches = h;
        rc = DSPFX_OK;
    }
    (void)hipFree(d);
    return rc;
}

extern "C" int dspfx_verify_libm(int device, int func, uint64_t *mismatches, uint32_t *max_ulp) {
    if (!mismatches || !max_ulp || func < 0 || func > 64) return DSPFX_ERR_INVALID;
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) return DSPFX_ERR_NO_DEVICE;
    if (device < 0 || device >= ndev || hipSetDevice(device) != hipSuccess) return DSPFX_ERR_INVALID;
    unsigned long long *d = nullptr, h[2] = {0, 0};
    if (hipMalloc((void **)&d, sizeof h) != hipSuccess) return DSPFX_ERR_OOM;
    int rc = DSPFX_ERR_HIP;
    if (hipMemset(d, 0, sizeof h) == hipSuccess && verify_libm_on_device(func, d, nullptr) == 0 &&
        hipMemcpy(h, d, sizeof h, hipMemcpyDeviceToHost) == hipSuccess) {
        *mismatches = h[0];
        *max_ulp = (uint32_t)std::min<unsigned long long>(h[1], 0xffffffffull);
        rc = DSPFX_OK;
    }
    (void)hipFree(d);
    return rc;
}

extern "C" int dspfx_profile_enable(dspfx_engine *e, int enable) {
    if (!e) return DSPFX_ERR_INVALID;
    ApiScope api(e);
    if (api.rc) return api.rc;
    e->profiling = enable != 0;
    if (enable > 0) {   // `enable` doubles as a hint: events for that many launches are created now,
        HIPCHK(e, hipSetDevice(e->device));   // outside the timed region (hipEventCreate is slow)
        while (e->ev_pool.size() < 2u * (size_t)enable * std::max<size_t>(1, e->stages.size())) {
            hipEvent_t ev = nullptr;
            HIPCHK(e, hipEventCreate(&ev));
            e->ev_pool.push_back(ev);
        }
    }
    return DSPFX_OK;
}

extern "C" int dspfx_profile_read(dspfx_engine *e, double *total_ms, uint32_t *launches, char *kernel_name,
                                  size_t cap, int reset) {
    if (!e) return DSPFX_ERR_INVALID;
    ApiScope api(e);
    if (api.rc) return api.rc;
    double best = -1.0;
    uint32_t best_n = 0;
    size_t best_stage = 0;
    for (size_t si = 0; si < e->prof.size(); ++si) {
        double tot = 0.0;
        for (auto &p : e->prof[si]) {
            HIPCHK(e, hipEventSynchronize(p.second));
            float ms = 0.0f;
            HIPCHK(e, hipEventElapsedTime(&ms, p.first, p.second));
            tot += ms;
        }
        if (tot > best) {
            best = tot;
            best_n = (uint32_t)e->prof[si].size();
            best_stage = si;
        }
    }
    if (total_ms) *total_ms = best < 0 ? 0.0 : best;
    if (launches) *launches = best_n;
    if (kernel_name && cap) {
        const char *nm = "";
        if (best_stage < e->stages.size()) {
            const Stage &st = e->stages[best_stage];
            nm = st.type == ST_FUSED ? (st.var ? st.var->name : "fused")
                 : st.type == ST_FUZZ ? "fuzz_kernel" : fir_kernel_name(e->nodes[(size_t)st.first].fir);
        }
        snprintf(kernel_name, cap, "%s", nm);
    }
    if (reset) {
        for (auto &st : e->prof) {
            for (auto &p : st) {
                e->ev_pool.push_back(p.first);
                e->ev_pool.push_back(p.second);
            }
            st.clear();
        }
    }
    return DSPFX_OK;
}

extern "C" double dspfx_algorithmic_bytes_per_sample(const dspfx_engine *e, uint32_t n_frames) {
    // 4 B in + 4 B out, + 8 B per delay line (tap read + write),
    // + per-block state traffic / n_frames, + 4 B side input for ADD/MIX,
    // FIR: + 4 B history write + 4*(T-1)/n_frames history re-read.
    if (!e || n_frames == 0) return 0.0;
    std::lock_guard<std::recursive_mutex> lk(e->api_mu);
    double b = 8.0;
    bool side = false;
    for (const Node &n : e->nodes) {
        switch (n.d.kind) {
        case DSPFX_BIQUAD: b += 32.0 / n_frames; break;
        case DSPFX_LOW_PASS:
        case DSPFX_HIGH_PASS:
        case DSPFX_SIGNAL_GEN:
        case DSPFX_ENVELOPE: b += 8.0 / n_frames; break;
        case DSPFX_REVERB: b += 8.0; break;
        case DSPFX_FIR: b += 4.0 + 4.0 * ((double)n.taps.size() - 1.0) / n_frames; break;
        case DSPFX_ADD:
        case DSPFX_MIX: side = true; break;
        default: break;
        }
    }
    if (side && !e->graph_mode) b += 4.0;   // a fused graph's "b" ports are fed from registers
    if (e->graph_mode) {                    // ... and a graph reads / writes exactly the blocks its links name
        unsigned in_mask = 0;
        int n_out = 1;
        for (const dspfx_graph_link &l : e->wiring) {
            if (graph_input_block(l.src) >= 0) in_mask |= 1u << graph_input_block(l.src);
            n_out = std::max(n_out, l.dst - (int)e->nodes.size() + 1);
        }
        b -= 4.0;                           // `in` was counted above
        b += 4.0 * __builtin_popcount(in_mask) + 4.0 * (n_out - 1);
    }
    return b;
}

extern "C" int dspfx_describe(const dspfx_engine *e, char *dst, size_t cap) {
    if (!e || !dst || cap == 0) return DSPFX_ERR_INVALID;
    std::lock_guard<std::recursive_mutex> lk(e->api_mu);
    static const char *kn[] = {"gain", "biquad", "low_pass", "high_pass", "reverb", "distort", "overdrive",
                               "chebyshev", "fir", "add", "mix", "signal_gen", "envelope"};
    std::string s;
    char buf[256];
    snprintf(buf, sizeof buf, "engine: N=%u max_frames=%u link_flags=%u\n", e->desc.channels, e->desc.max_frames,
             e->desc.link_flags);
    s += buf;
    for (size_t i = 0; i < e->stages.size(); ++i) {
        const Stage &st = e->stages[i];
        if (st.type == ST_FUSED) {
            snprintf(buf, sizeof buf, "stage %zu: fused kernel %s (F=%d, CPL=%d", i, st.var ? st.var->name : "?",
                     st.var ? st.var->f : 0, st.var ? st.var->cpl : 0);
            s += buf;
            if (st.var && !st.var->launch) {   // compiled at run time: say what the compiler allocated
                snprintf(buf, sizeof buf, ", %d VGPRs", reinterpret_cast<const JitKernel *>(st.var)->vgprs);
                s += buf;
            }
            if (st.var_ts) {
                snprintf(buf, sizeof buf, "; %d-frame blocks: time-sliced %s", 4 * st.var_ts->ts, st.var_ts->name);
                s += buf;
            }
            if (st.var_ts_tail) {
                snprintf(buf, sizeof buf, "; channels left over, %d-frame blocks: %s", 4 * st.var_ts_tail->ts, st.var_ts_tail->name);
                s += buf;
            }
            if (st.async) s += "; specialised kernels being compiled in the background, adopted at a block boundary (dspfx_kernels_ready)";
            if (st.async_mod) s += "; control-port kernel being compiled in the background";
            s += "):";
            for (int k = 0; k < st.count; ++k) {
                s += " ";
                s += kn[e->nodes[(size_t)(st.first + k)].d.kind];
            }
            if (st.count == 0 && i > 0 && e->stages[i - 1].type == ST_FIR)
                s += " (mix bus only; not launched for whole blocks: the FIR sweep leaves the bus' partial sums)";
            s += "\n";
        } else if (st.type == ST_FUZZ) {
            snprintf(buf, sizeof buf, "stage %zu: fuzz kernel\n", i);
            s += buf;
        } else {
            snprintf(buf, sizeof buf, "stage %zu: fir kernel %s (T=%zu)\n", i,
                     fir_kernel_name(e->nodes[(size_t)st.first].fir), e->nodes[(size_t)st.first].taps.size());
            s += buf;
        }
    }
    for (size_t i = 0; i < e->nodes.size(); ++i)
        if (!e->nodes[i].groups.empty()) {
            size_t reserved = 0;
            {
                std::lock_guard<std::mutex> lk(const_cast<dspfx_engine *>(e)->pool_mu);
                if (i < e->ring_pool.size()) reserved = e->ring_pool[i].size();
            }
            snprintf(buf, sizeof buf, "node %zu delay ring: %u samples in %zu of %zu groups x %zu MiB (+%zu reserved), %d re-placed by the placement probe\n", i,
                     e->nodes[i].D, ring_groups_for(e->nodes[i].D), e->nodes[i].groups.size(), (e->nodes[i].group_floats * sizeof(float)) >> 20, reserved,
                     e->nodes[i].ring_replaced);
            s += buf;
            if (getenv("DSPFX_DESCRIBE_GROUPS"))
                for (size_t g = 0; g < e->nodes[i].groups.size(); ++g) {
                    snprintf(buf, sizeof buf, "  group %zu @%p\n", g, (void *)e->nodes[i].groups[g]);
                    s += buf;
                }
        }
    {
        const std::string &cdir = e->env.cache_dir;
        snprintf(buf, sizeof buf, "run-time kernels of this process: %llu compiled, %llu loaded from the disk cache, %llu written to it (%s)\n",
                 (unsigned long long)g_jit_compiled.load(), (unsigned long long)g_jit_from_disk.load(), (unsigned long long)g_jit_disk_written.load(),
                 cdir.empty() ? "no disk cache" : cdir.c_str());
        s += buf;
        const std::string rej = jit_cache_rejected();
        if (!rej.empty()) {
            snprintf(buf, sizeof buf, "note: the disk cache %s is NOT used: the directory must belong to this user and be writable by nobody else "
                                      "(chmod go-w, or point DSPFX_CACHE_DIR elsewhere); kernels are recompiled by every process\n", rej.c_str());
            s += buf;
        }
    }
    if (e->jit_unavailable)
        s += "note: a run-time specialised kernel was wanted but could not be compiled (hiprtc unavailable, or DSPFX_KERNEL_HEADERS "
             "names a directory without chain_kernels.hip.h): the interpreting kernels serve, 7-25 % slower\n";
    snprintf(dst, cap, "%s", s.c_str());
    return DSPFX_OK;
}
