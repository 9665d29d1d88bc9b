// host_pipe.hip -- the host-buffer form of the process call (dspfx_process_host: pinned staging, upload / kernel / download of
// channel windows overlapped) and its allocator.  See engine.h for the split.
#include "engine.h"

using namespace dspfx;
using namespace dspfx_host;


extern "C" int dspfx_host_alloc(size_t bytes, void **out) {
    if (!out || bytes == 0) return DSPFX_ERR_INVALID;
    *out = nullptr;
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) return DSPFX_ERR_NO_DEVICE;
    if (hipHostMalloc(out, bytes, hipHostMallocDefault) != hipSuccess) {
        (void)hipGetLastError();
        *out = nullptr;
        return DSPFX_ERR_OOM;
    }
    return DSPFX_OK;
}

extern "C" int dspfx_host_free(void *p) {
    if (!p) return DSPFX_OK;
    return hipHostFree(p) == hipSuccess ? DSPFX_OK : DSPFX_ERR_HIP;
}

namespace dspfx_host {
bool is_pinned_host(const void *p) {
    hipPointerAttribute_t at;
    memset(&at, 0, sizeof at);
    if (hipPointerGetAttributes(&at, p) != hipSuccess) {
        (void)hipGetLastError();
        return false;
    }
    return at.type == hipMemoryTypeHost;
}
}  // namespace dspfx_host

extern "C" int dspfx_process_host(dspfx_engine *e, const float *in, const float *side, float *out, float *mix,
                                  uint32_t n_frames) {
    if (!e) return DSPFX_ERR_INVALID;
    ApiScope api(e);
    if (api.rc) return api.rc;
    if (!in || !out) return fail(e, DSPFX_ERR_INVALID, "in/out must not be null");
    if (n_frames > e->desc.max_frames)
        return fail(e, DSPFX_ERR_INVALID, "n_frames %u > max_frames %u", n_frames, e->desc.max_frames);
    const size_t cap = (size_t)e->desc.max_frames * e->desc.channels * sizeof(float);
    const size_t bytes = (size_t)n_frames * e->desc.channels * sizeof(float);
    if (!e->h_in) HIPCHK(e, hipMalloc((void **)&e->h_in, cap));
    if (!e->h_out) HIPCHK(e, hipMalloc((void **)&e->h_out, cap));
    if (side && !e->h_side) HIPCHK(e, hipMalloc((void **)&e->h_side, cap));
    if (mix && !e->h_mix) HIPCHK(e, hipMalloc((void **)&e->h_mix, e->desc.max_frames * sizeof(float)));
    // Pipelined form: the block is cut into channel parts; while part p runs, part p+1 is uploaded and part p-1
    // downloaded (both directions of the bus busy).  Needs a single fused stage per part (no FIR / Fuzz / mix bus),
    // the frame-major layout and a block that is not split at a short delay line.
    bool fused_only = !e->desc.tile_channels && n_frames <= e->min_delay && !e->has_siggen && !e->collect_due && !e->mp_count;
    for (const Stage &st : e->stages) fused_only = fused_only && st.type == ST_FUSED;
    const uint32_t N = e->desc.channels;
    static const uint32_t part = getenv("DSPFX_HOST_PART") ? (uint32_t)atoi(getenv("DSPFX_HOST_PART")) : 65536u;   // channels per part (multiple of 1024); 32k 15.5, 64k 13.7, 128k 14.1, 256k 15.1 ms
    static const bool pipe_off = getenv("DSPFX_HOST_PIPELINE") && atoi(getenv("DSPFX_HOST_PIPELINE")) == 0;
    // page-locked buffers only (dspfx_host_alloc): copies from pageable memory are staged by the runtime and do not overlap
    if (fused_only && !pipe_off && N >= 2 * part && is_pinned_host(in) && is_pinned_host(out) && (!side || is_pinned_host(side))) {
        if (!e->hs_in) {
            HIPCHK(e, hipStreamCreateWithFlags(&e->hs_in, hipStreamNonBlocking));
            HIPCHK(e, hipStreamCreateWithFlags(&e->hs_out, hipStreamNonBlocking));
            HIPCHK(e, hipStreamCreateWithFlags(&e->hs_run, hipStreamNonBlocking));
        }
        {   // the parts run on the engine's own stream: order it behind whatever used the state last
            const int brc = bind_stream(e, e->hs_run);
            if (brc) return brc;
        }
        const uint32_t n_parts = (N + part - 1) / part;
        while (e->hev.size() < 2 * (size_t)n_parts) {
            hipEvent_t ev = nullptr;
            HIPCHK(e, hipEventCreateWithFlags(&ev, hipEventDisableTiming));
            e->hev.push_back(ev);
        }
        const size_t pitch = (size_t)N * sizeof(float);
        int rc = DSPFX_OK;
        for (uint32_t p = 0; p < n_parts && rc == DSPFX_OK; ++p) {
            const uint32_t c0 = p * part, cn = std::min(part, N - c0);
            const size_t width = (size_t)cn * sizeof(float);
            HIPCHK(e, hipMemcpy2DAsync(e->h_in + c0, pitch, in + c0, pitch, width, n_frames, hipMemcpyHostToDevice, e->hs_in));
            if (side) HIPCHK(e, hipMemcpy2DAsync(e->h_side + c0, pitch, side + c0, pitch, width, n_frames, hipMemcpyHostToDevice, e->hs_in));
            HIPCHK(e, hipEventRecord(e->hev[2 * p], e->hs_in));
            HIPCHK(e, hipStreamWaitEvent(e->hs_run, e->hev[2 * p], 0));
            e->win_c0 = c0;
            e->win_n = cn;
            e->win_last = p + 1 == n_parts;
            if (mix) e->partials_override = e->mixpart;   // every part leaves its waves' partial sums; reduced once below
            rc = run_subblock(e, e->h_in, side ? e->h_side : nullptr, e->h_out, nullptr, n_frames, n_frames, e->hs_run);
            e->partials_override = nullptr;
            e->win_c0 = 0;
            e->win_n = 0;
            e->win_last = true;
            if (rc) break;
            HIPCHK(e, hipEventRecord(e->hev[2 * p + 1], e->hs_run));
            HIPCHK(e, hipStreamWaitEvent(e->hs_out, e->hev[2 * p + 1], 0));
            HIPCHK(e, hipMemcpy2DAsync(out + c0, pitch, e->h_out + c0, pitch, width, n_frames, hipMemcpyDeviceToHost, e->hs_out));
        }
        if (rc == DSPFX_OK && mix) {
            launch_mix_reduce(e->mixpart, e->mixpart_b, e->h_mix, n_frames, e->part_stride[e->flip], e->hs_run);
            HIPCHK(e, hipMemcpyAsync(mix, e->h_mix, n_frames * sizeof(float), hipMemcpyDeviceToHost, e->hs_run));
        }
        (void)hipStreamSynchronize(e->hs_in);
        (void)hipStreamSynchronize(e->hs_run);
        HIPCHK(e, hipStreamSynchronize(e->hs_out));
        if (rc == DSPFX_OK) e->frames_submitted += n_frames;
        return rc;
    }
    HIPCHK(e, hipMemcpy(e->h_in, in, bytes, hipMemcpyHostToDevice));
    if (side) HIPCHK(e, hipMemcpy(e->h_side, side, bytes, hipMemcpyHostToDevice));
    const int rc = dspfx_process(e, e->h_in, side ? e->h_side : nullptr, e->h_out, mix ? e->h_mix : nullptr,
                                 n_frames, nullptr);
    if (rc) return rc;
    HIPCHK(e, hipStreamSynchronize(nullptr));
    HIPCHK(e, hipMemcpy(out, e->h_out, bytes, hipMemcpyDeviceToHost));
    if (mix) HIPCHK(e, hipMemcpy(mix, e->h_mix, n_frames * sizeof(float), hipMemcpyDeviceToHost));
    return DSPFX_OK;
}

