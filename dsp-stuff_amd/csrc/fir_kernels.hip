// fir_kernels.hip -- see fir_kernels.h.  -ffp-contract=off like the rest of the library
// (the MFMA instruction is by definition a fused chain; the FIR bar is an RMS tolerance).
#include "fir_kernels.h"

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "../../include/dspfx.h"

namespace dspfx {

static thread_local std::string g_fir_err;
const char *fir_last_error() { return g_fir_err.c_str(); }

#define FIRCHK(call)                                                              \
    do {                                                                          \
        hipError_t e__ = (call);                                                  \
        if (e__ != hipSuccess) {                                                  \
            g_fir_err = std::string(#call) + ": " + hipGetErrorString(e__);       \
            return e__ == hipErrorOutOfMemory ? DSPFX_ERR_OOM : DSPFX_ERR_HIP;    \
        }                                                                         \
    } while (0)

constexpr int TILE_C = 32;   // channels per MFMA tile == ring tile width

__device__ __forceinline__ size_t ring_at(uint32_t c, uint32_t row, uint32_t R) {
    return ((size_t)(c >> 5) * R + row) * TILE_C + (c & 31);
}

// ring[(t0 + f) mod R] <- port value of in[f][c]  (fir.rs:193 push_back, after the
// collect_and_average hop when enabled).  blockIdx.y = group of 4 frames, x = channels: consecutive lanes take
// consecutive channels of one frame in both layouts (no 64-bit division per element).
constexpr uint32_t APPEND_FRAMES = 4;
__global__ void __launch_bounds__(256) fir_append_kernel(const float *in, float *ring, uint32_t N, uint32_t nframes,
                                                         uint32_t row0, uint32_t R, int hop, float hop_div,
                                                         const Layout lay) {
    const uint32_t c = blockIdx.x * 256 + threadIdx.x;
    if (c >= N) return;
    const uint32_t f0 = blockIdx.y * APPEND_FRAMES;
    float x[APPEND_FRAMES];
#pragma unroll
    for (uint32_t k = 0; k < APPEND_FRAMES; ++k)
        if (f0 + k < nframes) x[k] = __builtin_nontemporal_load(in + lay.at(f0 + k, c));
#pragma unroll
    for (uint32_t k = 0; k < APPEND_FRAMES; ++k) {
        if (f0 + k >= nframes) break;
        float v = x[k];
        if (hop) v = (0.0f + v) / hop_div;
        uint32_t r = row0 + f0 + k;
        r = r >= R ? r - R : r;
        ring[ring_at(c, r, R)] = v;
    }
}

// Exact path: one lane per (frame, channel); sequential f64 accumulation in deque
// order like Iterator::sum (fir.rs:204-206), cast to f32, + 0.0f (the empty `b`
// slice, 208-216), * divisor (222).  General in n0 (covers the warm-up quirk).
__global__ void __launch_bounds__(256) fir_exact_kernel(const float *ring, const double *taps, float *out, uint32_t N,
                                                        uint32_t nframes, uint32_t T, uint32_t R, uint64_t n0,
                                                        float divisor, const Layout lay) {
    const size_t total = (size_t)N * nframes;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
        const uint32_t f = (uint32_t)(i / N), c = (uint32_t)(i % N);
        const uint64_t n = n0 + f;                          // absolute index of this output
        const uint64_t first = n + 1 >= T ? n + 1 - T : 0;  // oldest sample still in the deque
        const uint32_t L = (uint32_t)(n - first + 1);       // deque length (<= T)
        uint32_t r = (uint32_t)(first % R);
        double acc = 0.0;
        for (uint32_t k = 0; k < L; ++k) {
            acc += (double)ring[ring_at(c, r, R)] * taps[k];
            r = r + 1 == R ? 0 : r + 1;
        }
        const float a = (float)acc;
        const float val = a + 0.0f;
        out[lay.at(f, c)] = val * divisor;
    }
}

// ---- MFMA path ------------------------------------------------------------------------
// One wave = one 32-channel tile x up to 128 output frames (4 MFMA tiles of 32 frames).
//   D[j][c] += W[j][k] * H[k][c]      A operand = W (lane: j = l&31, k = l>>5)
//                                     B operand = H (lane: c = l&31, k = l>>5)
//   C/D: lane holds column c = l&31, rows j = (r&3) + 8*(r>>2) + 4*(l>>5)  => each
//   accumulator register is one coalesced 128-byte output row segment.
// W[j][k] = taps_rev[idx],  idx = k - j in steady state (Toeplitz); while the deque is still
// filling (n < T-1) the reference pairs state[k] with taps[k] (fir.rs:204-206), i.e.
// idx = m = k + n0-T+1 and only samples m <= n exist: the WARM variant applies that map.
typedef float f32x16 __attribute__((ext_vector_type(16)));

struct FirMfmaArgs {
    const float *ring;
    const float *taps;     // [pad_lo + T + pad_hi], zeros in the pads
    float *out;
    uint32_t N, nframes, T, R;
    uint32_t rb;           // ring row holding time n0 - (T-1)   (k = 0)
    uint32_t kpad;         // K = T-1+nframes rounded up to a multiple of KC
    uint32_t pad_lo, pad_hi;
    long long n0;          // absolute index of the block's first output
    float divisor;
    int pad_;
    Layout lay;
};

constexpr int KC = 16;     // k per chunk (8 MFMA k-steps); history prefetched one chunk ahead
constexpr int FLUSH = 32;  // chunks per accumulator flush (512 terms)

template <bool WARM>
__global__ void __launch_bounds__(256) fir_mfma_kernel(const FirMfmaArgs a) {
    extern __shared__ float tp[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int ntp = (int)(a.pad_lo + a.T + a.pad_hi);
    for (int i = tid; i < ntp; i += 256) tp[i] = a.taps[i];
    __syncthreads();
    const uint32_t tile = blockIdx.x * 4 + wave;
    if ((size_t)tile * TILE_C >= a.N) return;
    const int cl = lane & 31, kh = lane >> 5;
    const float *hbase = a.ring + (size_t)tile * a.R * TILE_C + cl;

    // per output-tile weight index: LDS index = wofs[jt] + k   (k without the lane's kh, folded in)
    int wofs[4], whi[4];
#pragma unroll
    for (int jt = 0; jt < 4; ++jt) {
        const int j = jt * 32 + cl;
        if constexpr (WARM) {
            const long long first = a.n0 + j - (long long)a.T + 1;          // oldest sample of output j
            const int off = first >= 0 ? -j : (int)(a.n0 - (long long)a.T + 1);
            wofs[jt] = (int)a.pad_lo + kh + off;
            const long long hi = a.n0 + j < (long long)a.T - 1 ? a.n0 + j : (long long)a.T - 1;
            whi[jt] = (int)a.pad_lo + (int)hi;
        } else {
            wofs[jt] = (int)a.pad_lo + kh - j;
            whi[jt] = 0;
        }
    }

    f32x16 acc[4], tot[4];
#pragma unroll
    for (int jt = 0; jt < 4; ++jt)
#pragma unroll
        for (int r = 0; r < 16; ++r) { acc[jt][r] = 0.0f; tot[jt][r] = 0.0f; }

    // History rows of one chunk.  The chunk's first row is wave-uniform (scalar); unless the
    // chunk straddles the ring's wrap point (once per K sweep) every load is base + immediate.
    const float *hlane = hbase + (size_t)kh * TILE_C;
    auto load_chunk = [&](uint32_t kc, float (&h)[KC / 2]) {
        uint32_t row0 = a.rb + kc;                     // < 2R: rb < R, kc < kpad <= R + KC
        row0 = row0 >= a.R ? row0 - a.R : row0;
        if (row0 + KC + 1 <= a.R) {
            const float *p = hlane + (size_t)row0 * TILE_C;
#pragma unroll
            for (int s = 0; s < KC / 2; ++s) h[s] = __builtin_nontemporal_load(p + (size_t)(2 * s) * TILE_C);
        } else {
#pragma unroll
            for (int s = 0; s < KC / 2; ++s) {
                uint32_t row = row0 + 2 * s + kh;
                row = row >= a.R ? row - a.R : row;
                h[s] = __builtin_nontemporal_load(hbase + (size_t)row * TILE_C);
            }
        }
    };

    float h_cur[KC / 2], h_nxt[KC / 2];
    load_chunk(0, h_nxt);
    uint32_t kc = 0;
    while (kc < a.kpad) {
        // a fixed-trip inner sweep keeps the accumulators in place; its f32 chain is <= FLUSH*KC terms
        const uint32_t kend = kc + FLUSH * KC < a.kpad ? kc + FLUSH * KC : a.kpad;
        for (; kc < kend; kc += KC) {
#pragma unroll
            for (int s = 0; s < KC / 2; ++s) h_cur[s] = h_nxt[s];
            if (kc + KC < a.kpad) load_chunk(kc + KC, h_nxt);
#pragma unroll
            for (int s = 0; s < KC / 2; ++s) {
#pragma unroll
                for (int jt = 0; jt < 4; ++jt) {
                    const int idx = wofs[jt] + (int)kc + 2 * s;
                    float w;
                    if constexpr (WARM) {
                        const int lo = (int)a.pad_lo - 1;          // tp[pad_lo-1] == 0
                        const int ic = idx < lo ? lo : idx;
                        w = tp[ic];
                        w = idx <= whi[jt] ? w : 0.0f;              // samples newer than n do not exist yet
                    } else {
                        w = tp[idx];
                    }
                    acc[jt] = __builtin_amdgcn_mfma_f32_32x32x2f32(w, h_cur[s], acc[jt], 0, 0, 0);
                }
            }
        }
#pragma unroll
        for (int jt = 0; jt < 4; ++jt)
#pragma unroll
            for (int r = 0; r < 16; ++r) { tot[jt][r] = tot[jt][r] + acc[jt][r]; acc[jt][r] = 0.0f; }
    }
    const uint32_t c = tile * TILE_C + cl;
    if (c >= a.N) return;
#pragma unroll
    for (int jt = 0; jt < 4; ++jt)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const uint32_t j = jt * 32 + (r & 3) + 8 * (r >> 2) + 4 * kh;
            if (j < a.nframes) {
                const float val = tot[jt][r] + 0.0f;                   // fir.rs:216 `a + b` with the empty b slice
                __builtin_nontemporal_store(val * a.divisor, a.out + a.lay.at(j, c));   // fir.rs:222
            }
        }
}

int fir_configure(FirState &s, const double *taps_reversed, uint32_t n_taps, int mode, uint32_t N,
                  uint32_t max_frames) {
    fir_free(s);
    s.T = n_taps;
    s.N = N;
    s.max_frames = max_frames;
    s.mode = mode;
    s.R = n_taps - 1 + max_frames;
    if (s.R < 2 * KC) s.R = 2 * KC;                   // keeps the single wrap in load_chunk sufficient
    s.tiles = (N + TILE_C - 1) / TILE_C;
    s.pad_lo = 128;
    s.pad_hi = 128 + KC;
    s.n_seen = 0;
    const size_t ring_bytes = (size_t)s.tiles * s.R * TILE_C * sizeof(float);
    FIRCHK(hipMalloc((void **)&s.ring, ring_bytes));
    FIRCHK(hipMemset(s.ring, 0, ring_bytes));
    FIRCHK(hipMalloc((void **)&s.taps64, (size_t)n_taps * sizeof(double)));
    FIRCHK(hipMemcpy(s.taps64, taps_reversed, (size_t)n_taps * sizeof(double), hipMemcpyHostToDevice));
    std::vector<float> t32((size_t)s.pad_lo + n_taps + s.pad_hi, 0.0f);
    for (uint32_t i = 0; i < n_taps; ++i) t32[s.pad_lo + i] = (float)taps_reversed[i];
    FIRCHK(hipMalloc((void **)&s.taps32, t32.size() * sizeof(float)));
    FIRCHK(hipMemcpy(s.taps32, t32.data(), t32.size() * sizeof(float), hipMemcpyHostToDevice));
    // DSPFX_FIR_KERNEL: 0 = exact f64 VALU kernel, 1 = MFMA; default MFMA unless the filter is tiny
    const char *k = getenv("DSPFX_FIR_KERNEL");
    s.kernel = k ? atoi(k) : (n_taps >= 16 ? 1 : 0);
    const size_t lds = ((size_t)s.pad_lo + n_taps + s.pad_hi) * sizeof(float);
    if (lds > 160 * 1024 - 1024) s.kernel = 0;         // tap table must fit the CU's LDS
    if (s.kernel == 1 && lds > 64 * 1024) {
        FIRCHK(hipFuncSetAttribute((const void *)fir_mfma_kernel<false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        FIRCHK(hipFuncSetAttribute((const void *)fir_mfma_kernel<true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    }
    return 0;
}

void fir_free(FirState &s) {
    if (s.ring) (void)hipFree(s.ring);
    if (s.taps64) (void)hipFree(s.taps64);
    if (s.taps32) (void)hipFree(s.taps32);
    s.ring = nullptr;
    s.taps64 = nullptr;
    s.taps32 = nullptr;
}

void fir_reset(FirState &s) {
    if (s.ring) (void)hipMemset(s.ring, 0, (size_t)s.tiles * s.R * TILE_C * sizeof(float));
    s.n_seen = 0;
}

static unsigned grid_for(size_t total) {
    size_t b = (total + 255) / 256;
    if (b > 256 * 32) b = 256 * 32;
    return (unsigned)(b ? b : 1);
}

int fir_process(FirState &s, const float *in, float *out, uint32_t nframes, int hop, float hop_div,
                const Layout &lay, hipStream_t stream, hipEvent_t ev_begin, hipEvent_t ev_end) {
    if (nframes > s.max_frames) {
        g_fir_err = "nframes > max_frames";
        return DSPFX_ERR_INVALID;
    }
    const size_t total = (size_t)s.N * nframes;
    const uint32_t row0 = (uint32_t)(s.n_seen % s.R);
    hipLaunchKernelGGL(fir_append_kernel, dim3((s.N + 255) / 256, (nframes + APPEND_FRAMES - 1) / APPEND_FRAMES), dim3(256), 0,
                       stream, in, s.ring, s.N, nframes, row0, s.R, hop, hop_div, lay);
    // fir.rs:187-190
    const float divisor = s.mode == DSPFX_FIR_AVERAGE ? 1.0f / (float)s.T : 1.0f;
    if (ev_begin) (void)hipEventRecord(ev_begin, stream);
    if (s.kernel == 1) {
        // up to 128 output frames per launch (4 MFMA tiles); longer blocks go in slices
        for (uint32_t f0 = 0; f0 < nframes; f0 += 128) {
            const uint32_t nf = nframes - f0 < 128 ? nframes - f0 : 128;
            FirMfmaArgs a{};
            a.ring = s.ring;
            a.taps = s.taps32;
            a.out = out + (size_t)f0 * lay.ld;
            a.N = s.N;
            a.nframes = nf;
            a.T = s.T;
            a.R = s.R;
            a.n0 = (long long)(s.n_seen + f0);
            const long long t0 = a.n0 - (long long)s.T + 1;              // time of k = 0 (may be negative)
            a.rb = (uint32_t)(((t0 % (long long)s.R) + (long long)s.R) % (long long)s.R);
            const uint32_t K = s.T - 1 + nf;
            a.kpad = (K + KC - 1) / KC * KC;
            a.pad_lo = s.pad_lo;
            a.pad_hi = s.pad_hi;
            a.divisor = divisor;
            a.lay = lay;
            const unsigned grid = (s.tiles + 3) / 4;
            const size_t lds = ((size_t)s.pad_lo + s.T + s.pad_hi) * sizeof(float);
            if (a.n0 < (long long)s.T - 1)
                hipLaunchKernelGGL(fir_mfma_kernel<true>, dim3(grid), dim3(256), lds, stream, a);
            else
                hipLaunchKernelGGL(fir_mfma_kernel<false>, dim3(grid), dim3(256), lds, stream, a);
        }
    } else {
        hipLaunchKernelGGL(fir_exact_kernel, dim3(grid_for(total)), dim3(256), 0, stream, s.ring, s.taps64, out, s.N,
                           nframes, s.T, s.R, s.n_seen, divisor, lay);
    }
    if (ev_end) (void)hipEventRecord(ev_end, stream);
    FIRCHK(hipGetLastError());
    s.n_seen += nframes;
    return 0;
}

// exported state: u64 n_seen, then the T-1 most recent samples [t][N], oldest first
size_t fir_state_bytes(const FirState &s) { return 8 + (size_t)(s.T - 1) * s.N * sizeof(float); }

// one history row (time t) <-> a dense [N] host row: N/32 segments of 128 B, pitch R*128 B
static hipError_t copy_row(const FirState &s, uint32_t row, void *host, bool to_host) {
    const size_t seg = TILE_C * sizeof(float);
    const uint32_t full = s.N / TILE_C, rem = s.N % TILE_C;
    float *dev = s.ring + (size_t)row * TILE_C;
    hipError_t e = hipSuccess;
    if (full) {
        e = to_host ? hipMemcpy2D(host, seg, dev, (size_t)s.R * seg, seg, full, hipMemcpyDeviceToHost)
                    : hipMemcpy2D(dev, (size_t)s.R * seg, host, seg, seg, full, hipMemcpyHostToDevice);
        if (e != hipSuccess) return e;
    }
    if (rem) {
        float *d2 = dev + (size_t)full * s.R * TILE_C;
        char *h2 = (char *)host + (size_t)full * seg;
        e = to_host ? hipMemcpy(h2, d2, rem * sizeof(float), hipMemcpyDeviceToHost)
                    : hipMemcpy(d2, h2, rem * sizeof(float), hipMemcpyHostToDevice);
    }
    return e;
}

int fir_state_export(FirState &s, void *host_dst) {
    memcpy(host_dst, &s.n_seen, 8);
    char *dst = (char *)host_dst + 8;
    const size_t row = (size_t)s.N * sizeof(float);
    for (uint32_t k = 0; k + 1 < s.T; ++k) {
        // sample time n_seen - (T-1) + k ; before the start of time => zeros
        const int64_t t = (int64_t)s.n_seen - (int64_t)(s.T - 1) + k;
        if (t < 0) memset(dst + (size_t)k * row, 0, row);
        else FIRCHK(copy_row(s, (uint32_t)((uint64_t)t % s.R), dst + (size_t)k * row, true));
    }
    return 0;
}

int fir_state_import(FirState &s, const void *host_src) {
    uint64_t seen;
    memcpy(&seen, host_src, 8);
    const char *src = (const char *)host_src + 8;
    const size_t row = (size_t)s.N * sizeof(float);
    FIRCHK(hipMemset(s.ring, 0, (size_t)s.tiles * s.R * TILE_C * sizeof(float)));
    // re-base time so that the imported history occupies rows [0, hist)
    const uint64_t hist = seen < s.T - 1 ? seen : s.T - 1;
    for (uint64_t k = 0; k < hist; ++k) {
        const uint64_t srow = (s.T - 1) - hist + k;
        FIRCHK(copy_row(s, (uint32_t)k, (void *)(src + (size_t)srow * row), false));
    }
    s.n_seen = hist;
    return 0;
}

const char *fir_kernel_name(const FirState &s) { return s.kernel == 1 ? "fir_mfma_kernel" : "fir_exact_kernel"; }

}  // namespace dspfx
