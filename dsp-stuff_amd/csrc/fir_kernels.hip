// fir_kernels.hip -- see fir_kernels.h.  -ffp-contract=off like the rest of the library
// (the MFMA instruction is by definition a fused chain; the FIR bar is an RMS tolerance).
#include "fir_kernels.h"

#include <algorithm>
#include <cstdio>
#include <type_traits>
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
constexpr int KC = 16;       // k per chunk (8 MFMA k-steps); history prefetched one chunk ahead
constexpr int FLUSH = 32;    // chunks per accumulator flush (512 terms)
constexpr uint32_t SLICE = 128;          // output frames per launch (4 MFMA tiles of 32)
constexpr uint32_t PAD_LO = 160, PAD_HI = 160;   // zero pads of the LDS tap table: >= 127 + KC below, >= SLICE + KC above

__device__ __forceinline__ size_t ring_at(uint32_t c, uint32_t row, uint32_t R) {
    return ((size_t)(c >> 5) * R + row) * TILE_C + (c & 31);
}
__device__ __forceinline__ bool finite_f32(float v) { return __builtin_fabsf(v) < __builtin_inff(); }   // false for inf and NaN

// ring[(row0 + f) mod R] <- port value of in[f][c]  (fir.rs:193 push_back, after the collect_and_average hop when
// enabled).  Used when the MFMA kernel cannot append the block itself (warm-up, the delayed window after a tap reload,
// the exact kernel).  blockIdx.y = group of 4 frames, x = channels: consecutive lanes take consecutive channels of
// one frame in both layouts.  Non-finite samples raise the tile's flag.
constexpr uint32_t APPEND_FRAMES = 4;
__global__ void __launch_bounds__(256) fir_append_kernel(const float *in, float *ring, unsigned long long *nf_time, uint32_t N,
                                                         uint32_t nframes, uint32_t row0, uint32_t R, unsigned long long t0,
                                                         int hop, float hop_div, const Layout lay) {
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
        if (!finite_f32(v)) atomicMax(&nf_time[c >> 5], t0 + f0 + k + 1);
    }
}

// ---- exact path ---------------------------------------------------------------------------------------------
// One lane per (frame, channel); a workgroup = one 32-channel tile x 8 frames.  Output f of the slice sees the
// deque with front F = front0 + dfront[f] and n - F + 1 samples (n = n0 + f); its first physical slice holds
// n_a[f] of them (host-side model of std VecDeque).  fir.rs:204-216:
//   a = (sum_{k < min(n_a, T)} state[k] taps[k]) as f32          sequential f64, deque order
//   b = n_a < T ? (sum_{k < min(len - n_a, T - n_a)} state[n_a + k] taps[n_a + k]) as f32 : 0.0
//   out = (a + b) * divisor
// only_dirty: skip tiles without a non-finite sample at time >= t_lo (the fix-up pass behind the MFMA kernel).
struct FirExactArgs {
    const float *ring;
    const double *taps;
    float *out;
    const unsigned long long *nf_time;
    uint32_t N, nframes, T, R;
    unsigned long long n0, front0;
    long long t_lo;
    float divisor;
    int only_dirty;
    Layout lay;
    uint32_t n_a[SLICE];
    uint8_t dfront[SLICE];
};
__global__ void __launch_bounds__(256) fir_exact_kernel(const FirExactArgs a) {
    const uint32_t tile = blockIdx.x;
    if (a.only_dirty) {
        const unsigned long long lo = a.t_lo > 0 ? (unsigned long long)a.t_lo : 0ull;
        if (a.nf_time[tile] <= lo) return;
    }
    const uint32_t cl = threadIdx.x & 31, fi = threadIdx.x >> 5;
    const uint32_t c = tile * TILE_C + cl;
    if (c >= a.N) return;
    const float *col = a.ring + (size_t)tile * a.R * TILE_C + cl;
    // blockIdx.y strides over groups of 8 frames (the fix-up pass launches ONE block per tile: most exit above)
    for (uint32_t f = blockIdx.y * 8 + fi; f < a.nframes; f += gridDim.y * 8) {
        const unsigned long long F = a.front0 + a.dfront[f], n = a.n0 + f;
        const uint32_t len = (uint32_t)(n - F + 1), na = a.n_a[f];
        uint32_t r = (uint32_t)(F % a.R);
        const uint32_t la = na < a.T ? na : a.T;
        double acc = 0.0;
        for (uint32_t k = 0; k < la; ++k) {
            acc += (double)col[(size_t)r * TILE_C] * a.taps[k];
            r = r + 1 == a.R ? 0 : r + 1;
        }
        const float fa = (float)acc;
        float fb = 0.0f;
        if (na < a.T) {
            const uint32_t lb = (len - na) < (a.T - na) ? (len - na) : (a.T - na);
            double accb = 0.0;
            for (uint32_t k = 0; k < lb; ++k) {
                accb += (double)col[(size_t)r * TILE_C] * a.taps[na + k];
                r = r + 1 == a.R ? 0 : r + 1;
            }
            fb = (float)accb;
        }
        const float val = fa + fb;                                  // fir.rs:216
        a.out[a.lay.at(f, c)] = val * a.divisor;                    // fir.rs:222
    }
}

// ---- MFMA path ------------------------------------------------------------------------
// One wave = one 32-channel tile x up to 128 output frames (4 MFMA tiles of 32).
//   D[j][c] += W[j][k] * H[k][c]      A operand = W (lane: j = l&31, k = l>>5)
//                                     B operand = H (lane: c = l&31, k = l>>5)
//   C/D: lane holds column c = l&31, rows j = (r&3) + 8*(r>>2) + 4*(l>>5)  => each
//   accumulator register is one coalesced 128-byte output row segment.
// The sweep index k' runs over history rows from time t_k0 on; k = k' - koff is the row's distance from the oldest
// sample of output 0 (koff < KC pads the front so that the block's own samples start on a chunk boundary).
// W[j][k] = taps_rev[k - j] in steady state (Toeplitz).  While the deque is still filling the reference pairs
// state[m - front] with taps[m - front] (fir.rs:204-206) and only samples m <= n exist: the WARM variant applies
// that map.  FUSED (steady state, no extra delay): rows k' >= kring are the block itself -- read from `in`, hop
// applied, written to the ring, flagged when non-finite.
typedef float f32x16 __attribute__((ext_vector_type(16)));

struct FirMfmaArgs {
    float *ring;
    const float *taps;     // [pad_lo + T + pad_hi], zeros in the pads
    const float *in;       // FUSED: the slice's input block
    float *out;
    unsigned long long *nf_time;
    uint32_t N, nframes, T, R;
    uint32_t rb;           // ring row of k' = 0
    uint32_t kpad;         // sweep length, multiple of KC
    uint32_t kvalid;       // rows k' >= kvalid are not history (ring: not read)
    uint32_t kring;        // FUSED: first k' that comes from `in` (multiple of KC); otherwise kpad
    uint32_t koff;
    uint32_t row_new;      // FUSED: ring row that receives frame 0
    long long n0;          // absolute index of the slice's first output
    long long t_k0;        // absolute time of row k' = 0 (may be negative)
    long long tfront;      // WARM: absolute index of the deque's front
    float divisor;
    float hop_div;
    int hop;
    int pad_;
    double hop_rc;         // RN_f64(1 / hop_div)
    Layout lay;
};

// NJT = output tiles of 32 frames per wave.  4: one wave sweeps a channel tile's whole 128-frame block (every history row
// is loaded once; 240 registers = 2 waves per SIMD).  2: the block's two halves go to two waves of the SAME workgroup
// (the second read of a row hits in cache a few chunks later); half the accumulators = 4 waves per SIMD to cover each
// other's per-chunk bubbles, and a narrower Toeplitz band (K / T = 4159 / 4096 instead of 4223 / 4096).
template <bool WARM, bool FUSED, int NJT>
__global__ void __launch_bounds__(256) fir_mfma_kernel(const FirMfmaArgs a) {
    static_assert(NJT == 4 || (NJT == 2 && !FUSED), "the fused append needs the wave that sweeps the whole block");
    extern __shared__ float tp[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int ntp = (int)(PAD_LO + a.T + PAD_HI);
    for (int i = tid; i < ntp; i += 256) tp[i] = a.taps[i];
    __syncthreads();
    const uint32_t tile = NJT == 4 ? blockIdx.x * 4 + wave : blockIdx.x * 2 + (wave >> 1);
    const int j0 = NJT == 4 ? 0 : (wave & 1) * 64;        // first output frame of this wave
    if ((size_t)tile * TILE_C >= a.N || (uint32_t)j0 >= a.nframes) return;
    const int cl = lane & 31, kh = lane >> 5;
    float *hbase = a.ring + (size_t)tile * a.R * TILE_C + cl;
    const uint32_t c = tile * TILE_C + cl;
    const bool c_ok = c < a.N;
    // a non-finite sample somewhere in this tile's sweep: the ring loads are sanitised (wave-uniform)
    const bool dirty = a.nf_time[tile] > (unsigned long long)(a.t_k0 > 0 ? a.t_k0 : 0);

    // per output-tile weight index: LDS index = wofs[jt] + k'   (the lane's kh folded in)
    int wofs[NJT], whi[NJT];
#pragma unroll
    for (int jt = 0; jt < NJT; ++jt) {
        const int j = j0 + jt * 32 + cl;
        if constexpr (WARM) {
            const long long first = a.n0 + j - (long long)a.T + 1;            // front of output j's deque ...
            const long long Fj = first >= a.tfront ? first : a.tfront;        // ... which never moves before tfront
            wofs[jt] = (int)PAD_LO + kh + (int)(a.t_k0 - Fj);                 // idx = m - Fj, m = t_k0 + k'
            const long long hi = a.n0 + j - Fj < (long long)a.T - 1 ? a.n0 + j - Fj : (long long)a.T - 1;
            whi[jt] = (int)PAD_LO + (int)hi;                                  // samples newer than n do not exist yet
        } else {
            wofs[jt] = (int)PAD_LO + kh - (int)a.koff - j;
            whi[jt] = 0;
        }
    }

    f32x16 acc[NJT], tot[NJT];
#pragma unroll
    for (int jt = 0; jt < NJT; ++jt)
#pragma unroll
        for (int r = 0; r < 16; ++r) { acc[jt][r] = 0.0f; tot[jt][r] = 0.0f; }

    // History rows of one chunk.  The chunk's first row is wave-uniform (scalar); unless the chunk straddles the
    // ring's wrap point (once per sweep) every load is base + immediate.
    const float *hlane = hbase + (size_t)kh * TILE_C;
    const float *pin = nullptr;
    if constexpr (FUSED) pin = a.in + a.lay.at(0, c_ok ? c : 0);
    // load_chunk only ISSUES the loads of a chunk (one chunk ahead of its use); whatever has to look at the values --
    // the hop, the ring store and the non-finite check of the block's own rows, the sanitising of dirty history --
    // happens in `arrive`, when the chunk becomes the current one.  (Consuming a value right after its load -- or a
    // store between two loads, which the compiler must assume to alias -- serialises eight memory round trips per chunk:
    // measured +0.12 ms per block.)
    // NJT == 4 streams every history row exactly once: nontemporal.  NJT == 2 reads each row twice (the two halves of the
    // block, a few chunks apart): plain loads, so that the second read finds the line in L2.
    auto hload = [](const float *q) { return NJT == 4 ? __builtin_nontemporal_load(q) : *q; };
    auto load_chunk = [&](uint32_t kc, float (&h)[KC / 2]) {
        if (FUSED && kc >= a.kring) {
            // the block itself: frame f of the slice is row k' = kring + f
#pragma unroll
            for (int s = 0; s < KC / 2; ++s) {
                const uint32_t f = kc - a.kring + 2 * s + kh;
                h[s] = (f < a.nframes && c_ok) ? __builtin_nontemporal_load(pin + (size_t)f * a.lay.ld) : 0.0f;
            }
            return;
        }
        uint32_t row0 = a.rb + kc;                     // < 2R: rb < R, kc < kpad <= R
        row0 = row0 >= a.R ? row0 - a.R : row0;
        if (row0 + KC + 1 <= a.R && kc + KC <= a.kvalid) {
            const float *p = hlane + (size_t)row0 * TILE_C;
#pragma unroll
            for (int s = 0; s < KC / 2; ++s) h[s] = hload(p + (size_t)(2 * s) * TILE_C);
        } else {
#pragma unroll
            for (int s = 0; s < KC / 2; ++s) {
                uint32_t row = row0 + 2 * s + kh;
                row = row >= a.R ? row - a.R : row;
                h[s] = kc + 2 * s + kh < a.kvalid ? hload(hbase + (size_t)row * TILE_C) : 0.0f;
            }
        }
    };
    auto arrive = [&](uint32_t kc, float (&h)[KC / 2]) {
        if (FUSED && kc >= a.kring) {
            // branch-free per element (predicated store, selects): the hop as the exact product with RN_f64(1 / divisor) --
            // exact for 1.0001f like every divisor that is not an even integer (chain_kernels.hip.h, div_c)
            bool bad = false;
#pragma unroll
            for (int s = 0; s < KC / 2; ++s) {
                const uint32_t f = kc - a.kring + 2 * s + kh;
                const bool ok = c_ok && f < a.nframes;
                float v = h[s];
                if (a.hop) v = (float)((double)(0.0f + v) * a.hop_rc);        // node.rs:162-194, one pipe
                uint32_t row = a.row_new + f;
                row = row >= a.R ? row - a.R : row;
                if (ok) hbase[(size_t)row * TILE_C] = v;                      // fir.rs:193 push_back
                const bool fin = finite_f32(v);
                bad = bad || (ok && !fin);
                h[s] = (ok && fin) ? v : 0.0f;
            }
            if (__builtin_amdgcn_ballot_w64(bad)) {                           // rare: flag the tile for the exact fix-up pass
#pragma unroll
                for (int s = 0; s < KC / 2; ++s) {
                    const uint32_t f = kc - a.kring + 2 * s + kh;
                    if (c_ok && f < a.nframes) {
                        uint32_t row = a.row_new + f;
                        row = row >= a.R ? row - a.R : row;
                        if (!finite_f32(hbase[(size_t)row * TILE_C])) atomicMax(&a.nf_time[tile], (unsigned long long)(a.n0 + f + 1));
                    }
                }
            }
        } else if (dirty) {
#pragma unroll
            for (int s = 0; s < KC / 2; ++s) h[s] = finite_f32(h[s]) ? h[s] : 0.0f;
        }
    };

    // this wave's part of the sweep: outputs [j0, j0 + 32 NJT) have weights only for k in [j0, j0 + 32 NJT + T - 2]
    const uint32_t kc0 = (uint32_t)j0;                   // a multiple of KC
    uint32_t kc1 = (a.koff + (uint32_t)j0 + 32u * NJT + a.T - 1 + KC - 1) / KC * KC;
    kc1 = kc1 < a.kpad ? kc1 : a.kpad;
    float h_cur[KC / 2], h_nxt[KC / 2];
    load_chunk(kc0, h_nxt);

    // One chunk for the output tiles [LO, HI] (compile-time: the loop body holds exactly those MFMAs).
    auto chunk = [&](auto lo_c, auto hi_c, uint32_t kc) {
        constexpr int LO = decltype(lo_c)::value, HI = decltype(hi_c)::value;
#pragma unroll
        for (int s = 0; s < KC / 2; ++s) h_cur[s] = h_nxt[s];
        if (kc + KC < kc1) load_chunk(kc + KC, h_nxt);
        arrive(kc, h_cur);
#pragma unroll
        for (int s = 0; s < KC / 2; ++s) {
#pragma unroll
            for (int jt = LO; jt <= HI; ++jt) {
                const int idx = wofs[jt] + (int)kc + 2 * s;
                float w;
                if constexpr (WARM) {
                    const int lo = (int)PAD_LO - 1;          // tp[PAD_LO-1] == 0
                    const int ic = idx < lo ? lo : idx;
                    w = tp[ic];
                    w = idx <= whi[jt] ? w : 0.0f;
                } else {
                    w = tp[idx];
                }
                acc[jt] = __builtin_amdgcn_mfma_f32_32x32x2f32(w, h_cur[s], acc[jt], 0, 0, 0);
            }
        }
    };
    auto flush = [&]() {
#pragma unroll
        for (int jt = 0; jt < NJT; ++jt)
#pragma unroll
            for (int r = 0; r < 16; ++r) { tot[jt][r] = tot[jt][r] + acc[jt][r]; acc[jt][r] = 0.0f; }
    };
    // All four output tiles sweep the whole range: tile jt's weights are zero outside its Toeplitz band
    // [32 jt, 32 jt + 31 + T - 1], so its first 2 jt and last 6 - 2 jt chunks multiply zeros (2.3 % of the sweep at
    // T = 4096).  Skipping them per tile was tried two ways (a guarded second path; compile-time specialised segments):
    // both pushed the kernel past 256 registers (one wave per SIMD instead of two) and made the compiler shuttle the
    // accumulators AGPR <-> VGPR, costing far more than the corners.  A fixed-trip inner sweep keeps the accumulators in
    // place; its f32 chain is <= FLUSH * KC terms.
    using I0 = std::integral_constant<int, 0>;
    using IL = std::integral_constant<int, NJT - 1>;
    uint32_t kc = kc0;
    while (kc < kc1) {
        const uint32_t kend = kc + FLUSH * KC < kc1 ? kc + FLUSH * KC : kc1;
        for (; kc < kend; kc += KC) chunk(I0{}, IL{}, kc);
        flush();
    }
    if (!c_ok) return;
#pragma unroll
    for (int jt = 0; jt < NJT; ++jt)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const uint32_t j = j0 + jt * 32 + (r & 3) + 8 * (r >> 2) + 4 * kh;
            if (j < a.nframes) {
                const float val = tot[jt][r] + 0.0f;                   // fir.rs:216 `a + b` (one f32 sum here: the MFMA path's bar is an RMS tolerance)
                __builtin_nontemporal_store(val * a.divisor, a.out + a.lay.at(j, c));   // fir.rs:222
            }
        }
}

// old ring -> new ring for the sample times [t_begin, t_end): a tap reload that needs more rows
__global__ void __launch_bounds__(256) fir_rebase_kernel(const float *src, float *dst, uint32_t tiles, uint32_t R_src, uint32_t R_dst,
                                                         unsigned long long t_begin, unsigned long long t_end) {
    const size_t per_t = (size_t)tiles * TILE_C;
    const size_t total = (size_t)(t_end - t_begin) * per_t;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
        const unsigned long long t = t_begin + i / per_t;
        const size_t rem = i % per_t;
        const uint32_t tile = (uint32_t)(rem / TILE_C), cl = (uint32_t)(rem % TILE_C);
        dst[((size_t)tile * R_dst + (uint32_t)(t % R_dst)) * TILE_C + cl] = src[((size_t)tile * R_src + (uint32_t)(t % R_src)) * TILE_C + cl];
    }
}

static uint32_t ring_rows_for(uint64_t held, uint32_t n_taps, uint32_t max_frames) {
    // rows the sweep may touch: the deque (held samples, at least T-1), the block, the alignment pad and one chunk of slack
    uint64_t need = std::max<uint64_t>(held + 1, n_taps) + max_frames + 2 * KC;
    if (need < 4 * KC) need = 4 * KC;
    // an ODD row count: consecutive tiles are R * 128 bytes apart, and every wave of a launch walks its tile at about
    // the same row, so the tile stride decides how the concurrent 128-byte reads spread over the HBM channels.  Odd R
    // makes the stride an odd multiple of 128 B (measured: R = 4256 ran the 4096-tap sweep 3.4 % slower than R = 4223)
    return (uint32_t)need | 1u;
}

static int upload_taps(FirState &s, const double *taps_reversed, uint32_t n_taps) {
    if (s.taps64) (void)hipFree(s.taps64);
    if (s.taps32) (void)hipFree(s.taps32);
    s.taps64 = nullptr;
    s.taps32 = nullptr;
    s.T = n_taps;
    s.pad_lo = PAD_LO;
    s.pad_hi = PAD_HI;
    FIRCHK(hipMalloc((void **)&s.taps64, (size_t)n_taps * sizeof(double)));
    FIRCHK(hipMemcpy(s.taps64, taps_reversed, (size_t)n_taps * sizeof(double), hipMemcpyHostToDevice));
    std::vector<float> t32((size_t)PAD_LO + n_taps + PAD_HI, 0.0f);
    for (uint32_t i = 0; i < n_taps; ++i) t32[PAD_LO + i] = (float)taps_reversed[i];
    FIRCHK(hipMalloc((void **)&s.taps32, t32.size() * sizeof(float)));
    FIRCHK(hipMemcpy(s.taps32, t32.data(), t32.size() * sizeof(float), hipMemcpyHostToDevice));
    // DSPFX_FIR_KERNEL: 0 = exact f64 VALU kernel, 1 = MFMA; default MFMA unless the filter is tiny
    const char *k = getenv("DSPFX_FIR_KERNEL");
    s.kernel = k ? atoi(k) : (n_taps >= 16 ? 1 : 0);
    const size_t lds = ((size_t)PAD_LO + n_taps + PAD_HI) * sizeof(float);
    if (lds > 160 * 1024 - 1024) s.kernel = 0;         // tap table must fit the CU's LDS
    if (s.kernel == 1 && lds > 64 * 1024) {
        FIRCHK(hipFuncSetAttribute((const void *)fir_mfma_kernel<false, true, 4>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        FIRCHK(hipFuncSetAttribute((const void *)fir_mfma_kernel<false, false, 4>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        FIRCHK(hipFuncSetAttribute((const void *)fir_mfma_kernel<true, false, 4>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        FIRCHK(hipFuncSetAttribute((const void *)fir_mfma_kernel<false, false, 2>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        FIRCHK(hipFuncSetAttribute((const void *)fir_mfma_kernel<true, false, 2>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    }
    return 0;
}

int fir_configure(FirState &s, const double *taps_reversed, uint32_t n_taps, int mode, uint32_t N,
                  uint32_t max_frames) {
    fir_free(s);
    s.N = N;
    s.max_frames = max_frames;
    s.mode = mode;
    s.tiles = (N + TILE_C - 1) / TILE_C;
    s.n_seen = 0;
    s.front = 0;
    s.dq_cap = s.dq_head = 0;
    s.R = ring_rows_for(0, n_taps, max_frames);
    const size_t ring_bytes = (size_t)s.tiles * s.R * TILE_C * sizeof(float);
    FIRCHK(hipMalloc((void **)&s.ring, ring_bytes));
    FIRCHK(hipMemset(s.ring, 0, ring_bytes));
    FIRCHK(hipMalloc((void **)&s.nf_time, (size_t)s.tiles * sizeof(unsigned long long)));
    FIRCHK(hipMemset(s.nf_time, 0, (size_t)s.tiles * sizeof(unsigned long long)));
    return upload_taps(s, taps_reversed, n_taps);
}

int fir_set_taps(FirState &s, const double *taps_reversed, uint32_t n_taps, int mode) {
    if (!s.ring) {
        g_fir_err = "FIR node has no state yet";
        return DSPFX_ERR_STATE;
    }
    FIRCHK(hipDeviceSynchronize());                      // blocks in flight still read the old taps / ring
    s.mode = mode;
    const uint64_t held = s.n_seen - s.front;            // the deque survives the reload (fir.rs:153-171 touches `taps` only)
    const uint32_t need = ring_rows_for(held, n_taps, s.max_frames);
    if (need > s.R) {
        float *nr = nullptr;
        const size_t bytes = (size_t)s.tiles * need * TILE_C * sizeof(float);
        FIRCHK(hipMalloc((void **)&nr, bytes));
        FIRCHK(hipMemset(nr, 0, bytes));
        if (held) {
            hipLaunchKernelGGL(fir_rebase_kernel, dim3(1024), dim3(256), 0, nullptr, s.ring, nr, s.tiles, s.R, need, (unsigned long long)s.front,
                               (unsigned long long)s.n_seen);
            FIRCHK(hipGetLastError());
            FIRCHK(hipDeviceSynchronize());
        }
        (void)hipFree(s.ring);
        s.ring = nr;
        s.R = need;
    }
    return upload_taps(s, taps_reversed, n_taps);
}

void fir_free(FirState &s) {
    if (s.ring) (void)hipFree(s.ring);
    if (s.taps64) (void)hipFree(s.taps64);
    if (s.taps32) (void)hipFree(s.taps32);
    if (s.nf_time) (void)hipFree(s.nf_time);
    s.ring = nullptr;
    s.taps64 = nullptr;
    s.taps32 = nullptr;
    s.nf_time = nullptr;
}

void fir_reset(FirState &s) {
    if (s.ring) (void)hipMemset(s.ring, 0, (size_t)s.tiles * s.R * TILE_C * sizeof(float));
    if (s.nf_time) (void)hipMemset(s.nf_time, 0, (size_t)s.tiles * sizeof(unsigned long long));
    s.n_seen = 0;
    s.front = 0;
    s.dq_cap = s.dq_head = 0;
}

// One push_back (+ at most one pop_front, fir.rs:193-197) on the host-side model of std VecDeque<f64>: capacity grows
// by doubling from 4 and a wrapped ring moves its shorter part on growth (VecDeque::handle_capacity_increase).  Only the
// indices matter here.  Returns the length of the deque's first physical slice after the step (`a` of as_slices).
static uint32_t deque_step(FirState &s) {
    uint32_t len = (uint32_t)(s.n_seen - s.front);
    if (len == s.dq_cap) {
        const uint32_t old = s.dq_cap, ncap = old ? old * 2 : 4;
        if (s.dq_head > old - len) {                               // wrapped (len == old: any head != 0)
            const uint32_t head_len = old - s.dq_head, tail_len = len - head_len;
            if (!(head_len > tail_len && ncap - old >= tail_len)) s.dq_head = ncap - head_len;
        }
        s.dq_cap = ncap;
    }
    ++len;
    ++s.n_seen;
    if (len > s.T) {
        s.dq_head = s.dq_head + 1 == s.dq_cap ? 0 : s.dq_head + 1;
        --len;
        ++s.front;
    }
    return std::min(len, s.dq_cap - s.dq_head);
}

int fir_process(FirState &s, const float *in, float *out, uint32_t nframes, int hop, float hop_div,
                const Layout &lay, hipStream_t stream, hipEvent_t ev_begin, hipEvent_t ev_end) {
    if (nframes > s.max_frames) {
        g_fir_err = "nframes > max_frames";
        return DSPFX_ERR_INVALID;
    }
    // fir.rs:187-190
    const float divisor = s.mode == DSPFX_FIR_AVERAGE ? 1.0f / (float)s.T : 1.0f;
    // The MFMA kernel CAN append the block itself (FUSED: its newest rows come from `in`), which saves the append pass
    // (43 us at config 4) -- but measured on MI355X the fused kernel runs 0.10 ms longer (2.155 vs 2.056 ms per block,
    // profiles/r02_fir.txt), so the separate append pass stays the default; DSPFX_FIR_FUSE=1 selects the fused form.
    const char *fuse_env = getenv("DSPFX_FIR_FUSE");     // read per call (tests flip it)
    const bool no_fuse = !(fuse_env && atoi(fuse_env) == 1);
    bool ev_open = false;
    // up to 128 output frames per launch (4 MFMA tiles); longer blocks go in slices
    for (uint32_t f0 = 0; f0 < nframes; f0 += SLICE) {
        const uint32_t nf = std::min(SLICE, nframes - f0);
        const uint64_t n0 = s.n_seen, front0 = s.front;
        const uint64_t len0 = n0 - front0;
        const float *in_s = in + (size_t)f0 * lay.ld;
        float *out_s = out + (size_t)f0 * lay.ld;
        FirExactArgs ex{};
        for (uint32_t f = 0; f < nf; ++f) {                 // advances n_seen / front / the deque model
            ex.n_a[f] = deque_step(s);
            ex.dfront[f] = (uint8_t)(s.front - front0);
        }
        ex.ring = s.ring;
        ex.taps = s.taps64;
        ex.out = out_s;
        ex.nf_time = s.nf_time;
        ex.N = s.N;
        ex.nframes = nf;
        ex.T = s.T;
        ex.R = s.R;
        ex.n0 = n0;
        ex.front0 = front0;
        ex.t_lo = (long long)(front0 + ex.dfront[0]);
        ex.divisor = divisor;
        ex.lay = lay;
        const bool steady = len0 + 1 >= s.T;                 // the first output already sees T samples
        const uint64_t d = len0 > s.T ? len0 - s.T : 0;      // deque longer than the taps: a pure extra delay (fir.rs:195-197 pops one per step)
        const bool mfma = s.kernel == 1;
        const bool fused = mfma && steady && d == 0 && !no_fuse;
        s.last_fused = fused ? 1 : 0;
        if (!fused)
            hipLaunchKernelGGL(fir_append_kernel, dim3((s.N + 255) / 256, (nf + APPEND_FRAMES - 1) / APPEND_FRAMES), dim3(256), 0,
                               stream, in_s, s.ring, s.nf_time, s.N, nf, (uint32_t)(n0 % s.R), s.R, (unsigned long long)n0, hop, hop_div, lay);
        if (ev_begin && !ev_open) {
            (void)hipEventRecord(ev_begin, stream);
            ev_open = true;
        }
        const dim3 ex_grid(s.tiles, (nf + 7) / 8);
        if (mfma) {
            FirMfmaArgs a{};
            a.ring = s.ring;
            a.taps = s.taps32;
            a.in = in_s;
            a.out = out_s;
            a.nf_time = s.nf_time;
            a.N = s.N;
            a.nframes = nf;
            a.T = s.T;
            a.R = s.R;
            a.n0 = (long long)n0;
            a.koff = (KC - (s.T - 1) % KC) % KC;             // the block's own rows start on a chunk boundary
            a.t_k0 = (long long)n0 - (long long)d - (long long)(s.T - 1) - (long long)a.koff;   // may be negative
            a.rb = (uint32_t)(((a.t_k0 % (long long)s.R) + (long long)s.R) % (long long)s.R);
            a.kvalid = a.koff + s.T - 1 + nf;
            a.kpad = (a.kvalid + KC - 1) / KC * KC;
            a.kring = fused ? a.koff + s.T - 1 : a.kpad;
            a.row_new = (uint32_t)(n0 % s.R);
            a.tfront = (long long)front0;
            a.divisor = divisor;
            a.hop_div = hop_div;
            a.hop_rc = 1.0 / (double)hop_div;
            a.hop = hop;
            a.lay = lay;
            // two output tiles per wave (4 waves per SIMD, a narrower Toeplitz band) for short filters; DSPFX_FIR_NJT=2|4 forces either
            const char *njt_env = getenv("DSPFX_FIR_NJT");
            // (measured at 262144 channels, kernel ms: T = 256: 0.201 vs 0.214 with four tiles per wave; T = 1024: 0.611 vs 0.584;
            //  T = 4096: 2.257 vs 2.072 -- reading every history row twice costs more than the occupancy gains)
            const bool two = !fused && nf > 64 && (njt_env ? atoi(njt_env) == 2 : s.T <= 384);
            const unsigned grid = two ? (s.tiles + 1) / 2 : (s.tiles + 3) / 4;
            const size_t lds = ((size_t)PAD_LO + s.T + PAD_HI) * sizeof(float);
            if (!steady) {
                if (two) hipLaunchKernelGGL((fir_mfma_kernel<true, false, 2>), dim3(grid), dim3(256), lds, stream, a);
                else hipLaunchKernelGGL((fir_mfma_kernel<true, false, 4>), dim3(grid), dim3(256), lds, stream, a);
            } else if (fused) {
                hipLaunchKernelGGL((fir_mfma_kernel<false, true, 4>), dim3(grid), dim3(256), lds, stream, a);
            } else {
                if (two) hipLaunchKernelGGL((fir_mfma_kernel<false, false, 2>), dim3(grid), dim3(256), lds, stream, a);
                else hipLaunchKernelGGL((fir_mfma_kernel<false, false, 4>), dim3(grid), dim3(256), lds, stream, a);
            }
            if (ev_end && f0 + SLICE >= nframes) (void)hipEventRecord(ev_end, stream);
            // tiles holding a non-finite sample inside this slice's window are redone exactly (nothing to do otherwise)
            ex.only_dirty = 1;
            hipLaunchKernelGGL(fir_exact_kernel, dim3(s.tiles, 1), dim3(256), 0, stream, ex);
        } else {
            ex.only_dirty = 0;
            hipLaunchKernelGGL(fir_exact_kernel, ex_grid, dim3(256), 0, stream, ex);
            if (ev_end && f0 + SLICE >= nframes) (void)hipEventRecord(ev_end, stream);
        }
    }
    FIRCHK(hipGetLastError());
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
    FIRCHK(hipMemset(s.nf_time, 0, (size_t)s.tiles * sizeof(unsigned long long)));
    // re-base time so that the imported history occupies rows [0, hist): a deque of `hist` samples pushed from empty
    const uint64_t hist = seen < s.T - 1 ? seen : s.T - 1;
    for (uint64_t k = 0; k < hist; ++k) {
        const uint64_t srow = (s.T - 1) - hist + k;
        FIRCHK(copy_row(s, (uint32_t)k, (void *)(src + (size_t)srow * row), false));
    }
    s.n_seen = 0;
    s.front = 0;
    s.dq_cap = s.dq_head = 0;
    for (uint64_t k = 0; k < hist; ++k) (void)deque_step(s);
    return 0;
}

const char *fir_kernel_name(const FirState &s) { return s.kernel == 1 ? "fir_mfma_kernel" : "fir_exact_kernel"; }

}  // namespace dspfx
