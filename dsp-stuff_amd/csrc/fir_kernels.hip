// fir_kernels.hip -- see fir_kernels.h.  -ffp-contract=off like the rest of the library
// (the MFMA instruction is by definition a fused chain; the FIR bar is an RMS tolerance).
#include "fir_kernels.h"

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <type_traits>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "../../include/dspfx.h"
#include "aux_kernels.h"

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
constexpr int KC = 16;       // k per chunk: 8 k-steps of the f32 MFMA (K = 2), one bf16 MFMA (K = 16)
constexpr int FLUSH = 32;    // chunks per accumulator flush (about 512 terms per f32 chain)
constexpr uint32_t SLICE = 128;          // output frames per launch (4 MFMA tiles of 32)
constexpr uint32_t PAD_LO = 160, PAD_HI = 192;   // zero pads of the LDS tap table: >= 127 + KC below, >= SLICE + 3 KC above (weights are fetched one group ahead)

// History ring, chunk-transposed (R a multiple of KC = 16 rows; sample time t lives in row t mod R, so t mod 16 == row mod 16):
//     ring[tile][row / 16][(row / 4) % 2][(row / 8) % 2][channel in tile (32)][row % 4]          tile stride = R * 32 + 32 floats
// MFMA lane (c, kh) feeds one 16-row chunk with the eight CONSECUTIVE rows row0 + 8 kh ... row0 + 8 kh + 7 of channel c (K
// index of the lane's step s: 8 kh + s; the weights follow the same map).  Here those eight floats are two 16-byte
// pieces and a wave's chunk is one 2 KiB extent read by two fully coalesced dwordx4 loads per lane, whose only address
// arithmetic is the chunk's (scalar) offset -- instead of eight dword loads with a wrap test each: the loads and their
// address arithmetic were the largest non-MFMA cost of the row-major sweep (profiles/r02_fir.txt, experiment builds).
// The same eight consecutive rows are what one lane of the bf16 MFMAs (K = 16) of the split-precision sweep needs.
// A chunk never straddles the ring's wrap.  The tile stride gets one odd 128-byte pad so that concurrent waves, which
// walk their tiles at about the same row, spread over the HBM channels (R * 128 B alone would be a multiple of 2 KiB).
__host__ __device__ __forceinline__ size_t ring_tile_stride(uint32_t R) { return (size_t)R * TILE_C + TILE_C; }
__host__ __device__ __forceinline__ size_t ring_in_tile(uint32_t row, uint32_t cl) {
    return (size_t)(row >> 4) * (KC * TILE_C) + ((row >> 2) & 1) * 256 + ((row >> 3) & 1) * 128 + cl * 4 + (row & 3);
}
__host__ __device__ __forceinline__ size_t ring_at(uint32_t c, uint32_t row, uint32_t R) {
    return (size_t)(c >> 5) * ring_tile_stride(R) + ring_in_tile(row, c & 31);
}
static size_t ring_bytes_for(uint32_t tiles, uint32_t R) { return (size_t)tiles * ring_tile_stride(R) * sizeof(float); }
// false for inf, NaN -- and for finite samples of 2^127 and more, whose bf16 part would round to inf in the split-precision
// sweep: the MFMA sweeps treat all of them as zero and the tile is redone by the exact kernel
__device__ __forceinline__ bool finite_f32(float v) { return __builtin_fabsf(v) < 0x1p127f; }

// ring[(row0 + f) mod R] <- port value of in[f][c]  (fir.rs:193 push_back, after the collect_and_average hop when
// enabled).  One thread = one channel x four consecutive rows: the samples of one 16-byte piece of the ring.  Consecutive
// lanes take consecutive channels: the reads are coalesced 256-byte row segments in both I/O layouts, the writes 16 bytes
// per lane, contiguous over 32 lanes.  (Four channels x four rows per thread -- 16-byte loads, a 4 x 4 transpose in
// registers, 64 contiguous bytes of stores per lane -- was slower: every store instruction then touches a quarter of each line.)  blockIdx.y = piece index from the
// 16-row group that holds row0.  Non-finite samples raise the tile's flag.
__global__ void __launch_bounds__(256) fir_append_kernel(const float *in, float *ring, unsigned long long *nf_time, uint32_t N,
                                                         uint32_t nframes, uint32_t row0, uint32_t R, unsigned long long t0,
                                                         int hop, float hop_div, const Layout lay) {
    const uint32_t c = blockIdx.x * 256 + threadIdx.x;
    if (c >= N) return;
    const uint32_t piece = blockIdx.y, g = piece >> 2, q4 = piece & 3;      // q4: which four consecutive rows of the group
    // linear row (not yet wrapped) of the piece's first sample; frame f of the block is linear row row0 + f
    const uint32_t lin0 = (row0 & ~15u) + g * KC + q4 * 4;
    float x[4];
    bool ok[4];
#pragma unroll
    for (uint32_t k = 0; k < 4; ++k) {
        const int f = (int)(lin0 + k) - (int)row0;
        ok[k] = f >= 0 && (uint32_t)f < nframes;
        x[k] = ok[k] ? __builtin_nontemporal_load(in + lay.at((uint32_t)f, c)) : 0.0f;
    }
    uint32_t rg = (row0 & ~15u) + g * KC;                  // the group's first ring row: groups wrap as a whole
    rg = rg >= R ? rg - R : rg;
    float *dst = ring + ring_at(c, rg + q4 * 4, R);
#pragma unroll
    for (uint32_t k = 0; k < 4; ++k) {
        if (hop) x[k] = (0.0f + x[k]) / hop_div;
        if (ok[k] && !finite_f32(x[k])) atomicMax(&nf_time[c >> 5], t0 + (lin0 + k - row0) + 1);
    }
    if (ok[0] && ok[1] && ok[2] && ok[3]) {
        *(float4 *)dst = make_float4(x[0], x[1], x[2], x[3]);
    } else {
#pragma unroll
        for (uint32_t k = 0; k < 4; ++k)
            if (ok[k]) dst[k] = x[k];
    }
}
// pieces that can hold rows of a block of nframes starting anywhere in a 16-row group
static uint32_t append_pieces(uint32_t row0, uint32_t nframes) { return (((row0 & 15u) + nframes + KC - 1) / KC) * 4; }

// ---- the append pass of the two-part f16 sweep: f32 ring + packed {hi, lo} ring + per-epoch peaks ----------------------
// the sweep re-split every sample into f16 hi / lo and re-scanned it for its peak in EVERY block it stayed in
// the window -- 33 times at 4096 taps -- inside a kernel that runs at the socket's power limit.  The experiment build that fed the
// matrix pipe unsplit bits ran 15 % faster (0.873 -> 0.742 ms, 0.77 of the HBM peak: profiles/r05_fir_packed.txt).  So the
// split happens ONCE, here: beside the f32 ring (which the bf16 x 3 / exact passes, state export and tap reloads keep using) the
// block's samples go into a second ring already in the sweep's operand form (FirState::ringh), and each channel's peak over the
// 128 sample times of an "epoch" into a small table.  +4 bytes per sample for this pass, -(33 x the split) for the sweep.
// One thread = one channel x one epoch's part of the block (<= 128 consecutive sample times), walked in half chunks of 8 rows:
// the f32 ring gets two 16-byte pieces, the packed ring the lane's 16 bytes of hi parts and 16 bytes of lo parts per half chunk.
constexpr uint32_t EPOCH = 128;
__host__ __device__ __forceinline__ size_t ringh_at(uint32_t tile, uint32_t row8, uint32_t cl, uint32_t R) {   // dword index of the hi parts of rows row8 .. row8 + 7 (row8 % 8 == 0)
    return (size_t)tile * ring_tile_stride(R) + (size_t)(row8 >> 4) * (KC * TILE_C) + ((row8 >> 3) & 1) * 128 + cl * 4;
}
constexpr float HALF_APPEND_SCALE = 0x1p14f;
typedef float ap_f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned ap_u32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ unsigned pack_f16_pair(float a, float b) {     // v_cvt_pk_f16_f32: round to nearest even
    typedef _Float16 h2 __attribute__((ext_vector_type(2)));
    typedef float f2 __attribute__((ext_vector_type(2)));
    const h2 p = __builtin_convertvector(f2{a, b}, h2);
    unsigned u;
    __builtin_memcpy(&u, &p, 4);
    return u;
}
__device__ __forceinline__ float f16_lo_value(unsigned u) { return (float)__builtin_bit_cast(_Float16, (unsigned short)(u & 0xffffu)); }
__device__ __forceinline__ float f16_hi_value(unsigned u) { return (float)__builtin_bit_cast(_Float16, (unsigned short)(u >> 16)); }
// eight scaled samples (K order) -> hi[4], lo[4] dwords; non-finite / huge samples (already replaced by 0 by the caller) never get here
__device__ __forceinline__ void split8_packed(const float (&x)[8], unsigned (&hi)[4], unsigned (&lo)[4]) {
#pragma unroll
    for (int d = 0; d < 4; ++d) {
        const float a = x[2 * d] * HALF_APPEND_SCALE, b = x[2 * d + 1] * HALF_APPEND_SCALE;
        const unsigned h = pack_f16_pair(a, b);
        hi[d] = h;
        lo[d] = pack_f16_pair(a - f16_lo_value(h), b - f16_hi_value(h));      // exact differences (<= 13 significant bits)
    }
}
struct FirAppend2Args {
    const float *in;
    float *ring;
    unsigned *ringh;
    float *peaks;
    unsigned long long *nf_time;
    uint32_t N, nframes, R, peak_slots;
    unsigned long long n0;        // absolute time of the block's first frame
    int hop;
    float hop_div;
    Layout lay;
};
// A workgroup = 64 consecutive channels x four waves; wave w takes the epoch part's half chunks w, w + 4, w + 8, ... (so the four
// waves stream neighbouring rows at the same time and there are four times the loads in flight of one thread per channel: the
// first form of this kernel, one thread per channel walking all 128 rows, moved 4.6 TB/s); the four partial peaks meet in LDS.
constexpr int APPEND2_CH = 64;
__global__ void __launch_bounds__(256) fir_append2_kernel(const FirAppend2Args a) {
    __shared__ float part_peak[4][APPEND2_CH];
    const uint32_t w = threadIdx.x >> 6, ch = threadIdx.x & 63;
    const uint32_t c = blockIdx.x * APPEND2_CH + ch;
    const bool c_ok = c < a.N;
    const unsigned long long E = a.n0 / EPOCH + blockIdx.y;
    const unsigned long long t_end = a.n0 + a.nframes;
    const unsigned long long ta = E * EPOCH > a.n0 ? E * EPOCH : a.n0, tb = (E + 1) * EPOCH < t_end ? (E + 1) * EPOCH : t_end;
    const uint32_t tile = c >> 5, cl = c & 31;
    const float *__restrict__ in = a.in;
    float *__restrict__ ring = a.ring;
    unsigned *__restrict__ ringh = a.ringh;
    float peak = 0.0f;
    if (c_ok)
    for (unsigned long long h = (ta & ~7ull) + 8 * w; h < tb; h += 32) {
        float x[8];
        bool ok[8];
        bool all = true;
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            const unsigned long long t = h + k;
            ok[k] = t >= ta && t < tb;
            all = all && ok[k];
            x[k] = ok[k] ? __builtin_nontemporal_load(in + a.lay.at((uint32_t)(t - a.n0), c)) : 0.0f;
        }
        float xs[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            if (a.hop) x[k] = (0.0f + x[k]) / a.hop_div;
            const bool fin = finite_f32(x[k]);
            if (ok[k] && !fin) atomicMax(&a.nf_time[tile], h + k + 1);
            xs[k] = fin ? x[k] : 0.0f;                    // the packed ring never holds a non-finite part: such tiles are redone exactly
            if (ok[k]) peak = __builtin_fmaxf(peak, fin ? __builtin_fabsf(x[k]) * HALF_APPEND_SCALE : __builtin_inff());
        }
        const uint32_t r = (uint32_t)(h % a.R);          // R % 16 == 0 and h % 8 == 0: the eight rows share a chunk
        float *dst = ring + ring_at(c, r, a.R);          // rows r .. r+3, and r+4 .. r+7 one piece (256 floats) further
        unsigned hi[4], lo[4];
        split8_packed(xs, hi, lo);
        unsigned *ph = ringh + ringh_at(tile, r, cl, a.R);
        if (all) {
            __builtin_nontemporal_store(ap_f32x4{x[0], x[1], x[2], x[3]}, (ap_f32x4 *)dst);
            __builtin_nontemporal_store(ap_f32x4{x[4], x[5], x[6], x[7]}, (ap_f32x4 *)(dst + 256));
            __builtin_nontemporal_store(ap_u32x4{hi[0], hi[1], hi[2], hi[3]}, (ap_u32x4 *)ph);
            __builtin_nontemporal_store(ap_u32x4{lo[0], lo[1], lo[2], lo[3]}, (ap_u32x4 *)(ph + 256));
        } else {                                         // a block that starts or ends inside the half chunk: element by element
            unsigned short *ph16 = (unsigned short *)ph, *pl16 = (unsigned short *)(ph + 256);
#pragma unroll
            for (int k = 0; k < 8; ++k)
                if (ok[k]) {
                    dst[(k >> 2) * 256 + (k & 3)] = x[k];
                    ph16[k] = (unsigned short)(hi[k >> 1] >> (16 * (k & 1)));
                    pl16[k] = (unsigned short)(lo[k >> 1] >> (16 * (k & 1)));
                }
        }
    }
    part_peak[w][ch] = peak;
    __syncthreads();
    if (w != 0 || !c_ok || ta >= tb) return;
    peak = __builtin_fmaxf(__builtin_fmaxf(part_peak[0][ch], part_peak[1][ch]), __builtin_fmaxf(part_peak[2][ch], part_peak[3][ch]));
    // the epoch's peak: begun by the call that holds its first sample time, merged by the calls that continue it
    float *pk = a.peaks + ((size_t)tile * a.peak_slots + (uint32_t)(E % a.peak_slots)) * TILE_C + cl;
    if (E * EPOCH < a.n0) peak = __builtin_fmaxf(peak, *pk);
    *pk = peak;
}
// `ring` -> `ringh` + `peaks` for the sample times [t_lo, t_hi): after anything but the append pass wrote the f32 ring
__global__ void __launch_bounds__(256) fir_repack_kernel(const float *ring, unsigned *ringh, float *peaks, uint32_t N, uint32_t R, uint32_t peak_slots,
                                                         unsigned long long t_lo, unsigned long long t_hi) {
    const uint32_t c = blockIdx.x * 256 + threadIdx.x;
    if (c >= N) return;
    const uint32_t tile = c >> 5, cl = c & 31;
    for (unsigned long long E = t_lo / EPOCH + blockIdx.y; E * EPOCH < t_hi; E += gridDim.y) {
        const unsigned long long ta = E * EPOCH > t_lo ? E * EPOCH : t_lo, tb = (E + 1) * EPOCH < t_hi ? (E + 1) * EPOCH : t_hi;
        float peak = 0.0f;
        for (unsigned long long h = ta & ~7ull; h < tb; h += 8) {
            const uint32_t r = (uint32_t)(h % R);
            const float *src = ring + ring_at(c, r, R);
            float xs[8];
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                const float v = src[(k >> 2) * 256 + (k & 3)];
                const bool in = h + k >= ta && h + k < tb, fin = finite_f32(v);
                xs[k] = in && fin ? v : 0.0f;
                if (in) peak = __builtin_fmaxf(peak, fin ? __builtin_fabsf(v) * HALF_APPEND_SCALE : __builtin_inff());
            }
            unsigned hi[4], lo[4];
            split8_packed(xs, hi, lo);
            unsigned *ph = ringh + ringh_at(tile, r, cl, R);
            unsigned short *ph16 = (unsigned short *)ph, *pl16 = (unsigned short *)(ph + 256);
#pragma unroll
            for (int k = 0; k < 8; ++k)
                if (h + k >= ta && h + k < tb) {
                    ph16[k] = (unsigned short)(hi[k >> 1] >> (16 * (k & 1)));
                    pl16[k] = (unsigned short)(lo[k >> 1] >> (16 * (k & 1)));
                }
        }
        peaks[((size_t)tile * peak_slots + (uint32_t)(E % peak_slots)) * TILE_C + cl] = peak;
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
    uint32_t N, nframes, T, R, tiles;
    unsigned long long n0, front0;
    long long t_lo;
    float divisor;
    int only_dirty;
    float *mixpart;        // mix-bus partials [tiles][mix_ld] like the sweep's (fir_epilogue), or null
    uint32_t mix_ld;
    Layout lay;
    uint32_t n_a[SLICE];
    uint8_t dfront[SLICE];
};
__global__ void __launch_bounds__(256) fir_exact_kernel(const FirExactArgs a) {
  // (the fix-up pass behind a sweep launches a few hundred workgroups that walk the tiles and skip the clean ones: the flags of
  // up to 64 of a workgroup's tiles are fetched by ONE load -- lane j looks at tile base + j gridDim -- instead of one dependent
  // load per tile: the empty pass behind every block of config 4 took 6.6 us for 16 round trips)
  const unsigned long long lo = a.t_lo > 0 ? (unsigned long long)a.t_lo : 0ull;
  for (uint32_t base = blockIdx.x; base < a.tiles; base += gridDim.x * 64u) {
   const uint32_t mine = base + (threadIdx.x & 63u) * gridDim.x;
   unsigned long long todo = __ballot(mine < a.tiles && (!a.only_dirty || a.nf_time[mine] > lo));     // the same in every wave
   while (todo) {
    const uint32_t tile = base + (uint32_t)__builtin_ctzll(todo) * gridDim.x;
    todo &= todo - 1;
    const uint32_t cl = threadIdx.x & 31, fi = threadIdx.x >> 5;
    const uint32_t c = tile * TILE_C + cl;
    const bool c_ok = c < a.N;                 // (lanes past the last channel stay: they take part in the tile's sums)
    const float *col = a.ring + (size_t)tile * ring_tile_stride(a.R);
    // blockIdx.y strides over groups of 8 frames (one group per block in the fix-up pass: all of the tile in this block)
    for (uint32_t f = blockIdx.y * 8 + fi; f < a.nframes; f += gridDim.y * 8) {
        float o = 0.0f;
        if (c_ok) {
            const unsigned long long F = a.front0 + a.dfront[f], n = a.n0 + f;
            const uint32_t len = (uint32_t)(n - F + 1), na = a.n_a[f];
            uint32_t r = (uint32_t)(F % a.R);
            const uint32_t la = na < a.T ? na : a.T;
            double acc = 0.0;
            for (uint32_t k = 0; k < la; ++k) {
                acc += (double)col[ring_in_tile(r, cl)] * a.taps[k];
                r = r + 1 == a.R ? 0 : r + 1;
            }
            const float fa = (float)acc;
            float fb = 0.0f;
            if (na < a.T) {
                const uint32_t lb = (len - na) < (a.T - na) ? (len - na) : (a.T - na);
                double accb = 0.0;
                for (uint32_t k = 0; k < lb; ++k) {
                    accb += (double)col[ring_in_tile(r, cl)] * a.taps[na + k];
                    r = r + 1 == a.R ? 0 : r + 1;
                }
                fb = (float)accb;
            }
            const float val = fa + fb;                                  // fir.rs:216
            o = val * a.divisor;                                        // fir.rs:222
            a.out[a.lay.at(f, c)] = o;
        }
        if (a.mixpart) {                                                // the tile's sum over its channels for this frame
#pragma unroll
            for (int m = 16; m >= 1; m >>= 1) o = o + __shfl_xor(o, m, 32);
            if (cl == 0) a.mixpart[(size_t)tile * a.mix_ld + f] = o;
        }
    }
   }
  }
}

// ---- fill phase of a deque that started empty ------------------------------------------------------------------
// fir.rs:193-214 with front == 0 and no pop yet: out[n] = (sum_{m <= n} state[m] * taps[m]) as f32 + 0.0, times the divisor.
// The reference forms that sum afresh for every output, sequentially in f64 from m = 0: exactly the running sum kept
// here per channel, so the outputs are the reference's bit for bit.  One thread per channel walks the slice's frames; the
// 32 channels of a tile are the lanes of one half-wave (mix-bus partials like fir_exact_kernel's).
__global__ void __launch_bounds__(256) fir_warm_scan_kernel(const float *in, float *out, double *acc, const double *taps, uint32_t N,
                                                            uint32_t nframes, uint32_t n0, int hop, float hop_div, float divisor,
                                                            float *mixpart, uint32_t mix_ld, const Layout lay) {
    const uint32_t c = blockIdx.x * 256 + threadIdx.x;
    const bool c_ok = c < N;
    double a = c_ok ? acc[c] : 0.0;
    for (uint32_t f = 0; f < nframes; ++f) {
        float o = 0.0f;
        if (c_ok) {
            float x = __builtin_nontemporal_load(in + lay.at(f, c));
            if (hop) x = (0.0f + x) / hop_div;                         // node.rs:162-194, one pipe (as in fir_append_kernel)
            a += (double)x * taps[n0 + f];                             // fir.rs:204-206
            o = ((float)a + 0.0f) * divisor;                           // fir.rs:216 (`b` is the empty second slice), 222
            __builtin_nontemporal_store(o, out + lay.at(f, c));
        }
        if (mixpart) {
#pragma unroll
            for (int m = 16; m >= 1; m >>= 1) o = o + __shfl_xor(o, m, 32);
            if ((threadIdx.x & 31) == 0 && c_ok) mixpart[(size_t)(c >> 5) * mix_ld + f] = o;
        }
    }
    if (c_ok) acc[c] = a;
}

// ---- MFMA path ------------------------------------------------------------------------
// One wave = one 32-channel tile x up to 128 output frames (4 MFMA tiles of 32).
//   D[j][c] += W[j][k] * H[k][c]      A operand = W (lane: j = l&31, k = l>>5)
//                                     B operand = H (lane: c = l&31, k = l>>5)
//   C/D: lane holds column c = l&31, rows j = (r&3) + 8*(r>>2) + 4*(l>>5)  => each
//   accumulator register is one coalesced 128-byte output row segment.
// The sweep index k' runs over history rows from time t_k0 on, t_k0 a multiple of KC (so every chunk is one 16-row
// group of the ring); k = k' - koff is the row's distance from the oldest sample of output 0 (koff < KC).
// W[j][k] = taps_rev[k - j] in steady state (Toeplitz).  While the deque is still filling the reference pairs
// state[m - front] with taps[m - front] (fir.rs:204-206) and only samples m <= n exist: the WARM variant applies
// that map.
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

struct FirMfmaArgs {
    const float *ring;
    const unsigned *ringh; // two-part f16 sweep: the packed {hi, lo} history (FirState::ringh), or null: split the f32 history in the sweep
    const float *peaks;    // ... and the per-epoch channel peaks [tiles][peak_slots][32]
    uint32_t peak_slots;
    const float *taps;     // [pad_lo + T + pad_hi], zeros in the pads
    const unsigned *taps_split;   // split-precision sweep: [3][ntp4] bf16 pair tables (entry m: parts of taps m, m + 1)
    const unsigned *taps_half;    // two-part f16 sweep: [2][ntp4] f16 pair tables of the taps scaled by a power of two
    float half_unscale;           // ... and what an accumulator of that sweep is multiplied by (exact: a power of two)
    unsigned *redo_count;         // two-part sweep: tiles whose window it cannot serve are listed here (count, then the tiles) ...
    unsigned *redo_tiles;
    unsigned *redo_clear;         // ... and the NEXT block's count is zeroed by this launch
    const unsigned *only_count;   // bf16 x 3 sweep as the two-part sweep's second pass: only the tiles listed (null: all tiles)
    const unsigned *only_tiles;
    float *out;
    const unsigned long long *nf_time;
    uint32_t N, nframes, T, R;
    uint32_t rb;           // ring row of k' = 0 (a multiple of KC)
    uint32_t kpad;         // sweep length, multiple of KC
    uint32_t kvalid;       // rows k' >= kvalid are not history (stale ring rows: masked)
    uint32_t koff;
    long long n0;          // absolute index of the slice's first output
    long long t_k0;        // absolute time of row k' = 0 (may be negative)
    long long tfront;      // WARM: absolute index of the deque's front
    float divisor;
    float *mixpart;        // mix-bus partials [tiles][mix_ld] (this slice's first frame at column 0), or null
    uint32_t mix_ld;
    FirMixPipe mp;         // earlier blocks' mix-bus stages hosted by this launch's first workgroups (stage == 0: none)
    Layout lay;
};
__device__ __forceinline__ void fir_mixpipe_prologue(const FirMfmaArgs &a) {      // = mixpipe_prologue of the chain kernels
    const unsigned b = blockIdx.x;
    if ((a.mp.stage & 1) && b < MIX_SLICES) mix_slice_reduce(a.mp.prev_a, a.mp.cur_b, a.mp.rows_a, a.mix_ld, b, threadIdx.x, blockDim.x);
    else if ((a.mp.stage & 2) && b == MIX_SLICES) mix_final_reduce(a.mp.prev_b, a.mp.mix, a.mix_ld, a.mp.div, threadIdx.x, blockDim.x);
}

// The end of a sweep: out = (a + b) * divisor (fir.rs:216-222), each accumulator register one coalesced 128-byte row of the
// output -- and, when the FIR node ends the chain, the Output node's mix bus: the tile's per-frame sums over its 32 channels
// (a reduce-scatter over the 32 lanes of each half-wave: 62 cross-lane adds for the 64 frames a half holds), one row of
// partials per tile, reduced further by the engine's usual slice / final stages.  Without this the engine ran an
// empty chain kernel over the FIR output just to sum it (32 us per block at config 4).
template <int NJT, class Get>
__device__ __forceinline__ void fir_epilogue(const FirMfmaArgs &a, uint32_t tile, uint32_t c, bool c_ok, int j0, int kh, int lane, Get get) {
    constexpr int NV = NJT * 16;
    float o[NV];
#pragma unroll
    for (int jt = 0; jt < NJT; ++jt)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const float val = get(jt, r) + 0.0f;                       // fir.rs:216 `a + b` (one f32 sum here: the MFMA path's bar is an RMS tolerance)
            o[jt * 16 + r] = val * a.divisor;                          // fir.rs:222
            const uint32_t j = j0 + jt * 32 + (r & 3) + 8 * (r >> 2) + 4 * kh;
            if (c_ok && j < a.nframes) __builtin_nontemporal_store(o[jt * 16 + r], a.out + a.lay.at(j, c));
        }
    if (!a.mixpart) return;
#pragma unroll
    for (int i = 0; i < NV; ++i) o[i] = c_ok ? o[i] : 0.0f;
    rs_stage<NV, NV, 16>(o, lane);                 // lane (l & 31) of each half now holds the sums of values (NV/32) (l & 31) + i
#pragma unroll
    for (int i = 0; i < NV / 32; ++i) {
        const int idx = (NV / 32) * (lane & 31) + i, jt = idx >> 4, r = idx & 15;
        const uint32_t j = j0 + jt * 32 + (r & 3) + 8 * (r >> 2) + 4 * kh;
        if (j < a.nframes) a.mixpart[(size_t)tile * a.mix_ld + j] = o[i];
    }
}

// NJT = output tiles of 32 frames per wave.  4: one wave sweeps a channel tile's whole 128-frame block (every history row
// is loaded once; 240 registers = 2 waves per SIMD).  2: the block's two halves go to two waves of the SAME workgroup
// (the second read of a row hits in cache a few chunks later); half the accumulators = 4 waves per SIMD to cover each
// other's per-chunk bubbles, and a narrower Toeplitz band (K / T = 4159 / 4096 instead of 4223 / 4096).
template <bool WARM, int NJT>
__global__ void __launch_bounds__(256) fir_mfma_kernel(const FirMfmaArgs a) {
    extern __shared__ float tp[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int ntp = (int)(PAD_LO + a.T + PAD_HI);
    for (int i = tid; i < ntp; i += 256) tp[i] = a.taps[i];
    __syncthreads();
    const uint32_t tile = NJT == 4 ? blockIdx.x * 4 + wave : blockIdx.x * 2 + (wave >> 1);
    const int j0 = NJT == 4 ? 0 : (wave & 1) * 64;        // first output frame of this wave
    if ((size_t)tile * TILE_C >= a.N || (uint32_t)j0 >= a.nframes) return;
    const int cl = lane & 31, kh = lane >> 5;
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
            wofs[jt] = (int)PAD_LO + 8 * kh + (int)(a.t_k0 - Fj);             // idx = m - Fj, m = t_k0 + k'
            const long long hi = a.n0 + j - Fj < (long long)a.T - 1 ? a.n0 + j - Fj : (long long)a.T - 1;
            whi[jt] = (int)PAD_LO + (int)hi;                                  // samples newer than n do not exist yet
        } else {
            wofs[jt] = (int)PAD_LO + 8 * kh - (int)a.koff - j;
            whi[jt] = 0;
        }
    }

    f32x16 acc[NJT], tot[NJT];
#pragma unroll
    for (int jt = 0; jt < NJT; ++jt)
#pragma unroll
        for (int r = 0; r < 16; ++r) { acc[jt][r] = 0.0f; tot[jt][r] = 0.0f; }

    // History rows of one chunk: the lane's two 16-byte pieces of the chunk's 2 KiB extent (ring layout above); the
    // chunk's offset is wave-uniform.  load_chunk only ISSUES the loads (one chunk ahead of their use); whatever has to
    // look at the values -- masking the rows past the newest sample, sanitising dirty history -- happens in `arrive`,
    // when the chunk becomes the current one.
    // NJT == 4 streams every history row exactly once: nontemporal.  NJT == 2 reads each row twice (the two halves of the
    // block, a few chunks apart): plain loads, so that the second read finds the line in L2.
    const float *hlane = a.ring + (size_t)tile * ring_tile_stride(a.R) + (size_t)(kh * 32 + cl) * 4;
    auto load_chunk = [&](uint32_t kc, f32x4 (&h)[2]) {
        uint32_t row0 = a.rb + kc;                     // < 2R: rb < R, kc < kpad <= R; a multiple of KC
        row0 = row0 >= a.R ? row0 - a.R : row0;
        const f32x4 *p = (const f32x4 *)(hlane + (size_t)row0 * TILE_C);
        if constexpr (NJT == 4) {
            h[0] = __builtin_nontemporal_load(p);
            h[1] = __builtin_nontemporal_load(p + 64);
        } else {
            h[0] = p[0];
            h[1] = p[64];
        }
    };
    auto arrive = [&](uint32_t kc, float (&h)[KC / 2]) {
        if (kc + KC > a.kvalid) {                      // the sweep's last chunk: rows past the block's newest sample are stale
#pragma unroll
            for (int s = 0; s < KC / 2; ++s) h[s] = kc + 8 * kh + s < a.kvalid ? h[s] : 0.0f;
        }
        if (dirty) {
#pragma unroll
            for (int s = 0; s < KC / 2; ++s) h[s] = finite_f32(h[s]) ? h[s] : 0.0f;
        }
    };

    // this wave's part of the sweep: outputs [j0, j0 + 32 NJT) have weights only for k in [j0, j0 + 32 NJT + T - 2]
    const uint32_t kc0 = (uint32_t)j0;                   // a multiple of KC
    uint32_t kc1 = (a.koff + (uint32_t)j0 + 32u * NJT + a.T - 1 + KC - 1) / KC * KC;
    kc1 = kc1 < a.kpad ? kc1 : a.kpad;
    float h_cur[KC / 2];
    f32x4 h_nxt[2];
    load_chunk(kc0, h_nxt);

    // Weights of group g of chunk kc: the chunk's eight k-steps go in four groups of two (2 NJT MFMAs each).  The LDS
    // reads of a group are issued one group ahead of their MFMAs -- the last group of a chunk fetches the first group of
    // the NEXT chunk (the table's upper pad covers the read past the sweep's end) -- so that no MFMA waits for an LDS
    // round trip issued just before it.
    auto wload = [&](uint32_t kc, int g, float (&w)[NJT][2]) {
#pragma unroll
        for (int jt = 0; jt < NJT; ++jt)
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                const int idx = wofs[jt] + (int)kc + 2 * g + u;
                if constexpr (WARM) {
                    const int lo = (int)PAD_LO - 1;          // tp[PAD_LO-1] == 0
                    w[jt][u] = tp[(idx < lo || idx > whi[jt]) ? lo : idx];
                } else {
                    w[jt][u] = tp[idx];
                }
            }
    };
    float wq[2][NJT][2];
    wload(kc0, 0, wq[0]);
    auto chunk = [&](uint32_t kc) {
#pragma unroll
        for (int s = 0; s < KC / 2; ++s) h_cur[s] = h_nxt[s >> 2][s & 3];
        if (kc + KC < kc1) load_chunk(kc + KC, h_nxt);
        arrive(kc, h_cur);
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            if (g < 3) wload(kc, g + 1, wq[(g + 1) & 1]);
            else wload(kc + KC, 0, wq[0]);
#pragma unroll
            for (int u = 0; u < 2; ++u)
#pragma unroll
                for (int jt = 0; jt < NJT; ++jt)
                    acc[jt] = __builtin_amdgcn_mfma_f32_32x32x2f32(wq[g & 1][jt][u], h_cur[2 * g + u], acc[jt], 0, 0, 0);
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
    uint32_t kc = kc0;
    while (kc < kc1) {
        const uint32_t kend = kc + FLUSH * KC < kc1 ? kc + FLUSH * KC : kc1;
        for (; kc < kend; kc += KC) chunk(kc);
        flush();
    }
    fir_epilogue<NJT>(a, tile, c, c_ok, j0, kh, lane, [&](int jt, int r) { return tot[jt][r]; });
}

// ---- steady state, skewed sweep -----------------------------------------------------------------------------
// In steady state W is Toeplitz: W[j][k] = taps_rev[k - j].  Output tile jt (frames 32 jt ...) times history chunk
// i + 2 jt needs the weights taps_rev[16 i + kk - jj] -- the SAME for every jt.  So iteration i of this kernel gives all
// NJT output tiles one shared set of weights (8 LDS values per lane instead of 8 NJT) and tile jt its own history chunk
// i + 2 jt, kept in a register window of 2 NJT - 1 + D chunks (the ones in use plus D requested ahead; slot = chunk mod
// window; the loop is unrolled by the window so that every slot is a fixed set of registers).  Every tile then sweeps exactly its own band: (T + 30 + koff) / 16 + 1
// iterations instead of the (T + 32 NJT - 2 + koff) / 16 + 1 chunks of the rectangular sweep above, whose tiles
// multiply the zero corners of the band (2.3 % of the MFMAs at T = 4096, NJT = 4).
template <int P, int N, class F>
__device__ __forceinline__ void static_for(F &&f) {
    if constexpr (P < N) {
        f(std::integral_constant<int, P>{});
        static_for<P + 1, N>(f);
    }
}

template <int NJT, int D>
__global__ void __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(NJT == 4 ? 2 : 4))) fir_skew_kernel(const FirMfmaArgs a) {
    // WIN chunks are in use by the NJT tiles of one iteration (every second one; the others belong to the next iteration),
    // D more are in flight: a chunk is requested D iterations before its first use
    constexpr int WIN = 2 * (NJT - 1) + 1, SLOTS = WIN + D;
    constexpr int FLUSH_S = SLOTS * ((FLUSH + SLOTS / 2) / SLOTS);      // iterations per flush: a whole number of unrolled bodies
    static_assert(SLOTS % 2 == 0, "the weight registers alternate with the iteration's parity");
    if (a.mp.stage) fir_mixpipe_prologue(a);
    extern __shared__ float tp[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int ntp = (int)(PAD_LO + a.T + PAD_HI);
    for (int i = tid; i < ntp; i += 256) tp[i] = a.taps[i];
    __syncthreads();
    const uint32_t tile = NJT == 4 ? blockIdx.x * 4 + wave : blockIdx.x * 2 + (wave >> 1);
    const int j0 = NJT == 4 ? 0 : (wave & 1) * 64;        // first output frame of this wave
    if ((size_t)tile * TILE_C >= a.N || (uint32_t)j0 >= a.nframes) return;
    const int cl = lane & 31, kh = lane >> 5;
    const uint32_t c = tile * TILE_C + cl;
    const bool c_ok = c < a.N;
    const bool dirty = a.nf_time[tile] > (unsigned long long)(a.t_k0 > 0 ? a.t_k0 : 0);
    const uint32_t cb = (uint32_t)j0 / KC;                // this wave's chunk 0 in sweep chunks
    const int wofs = (int)PAD_LO + 8 * kh - (int)a.koff - cl;  // LDS index of (iteration i, step s) = wofs + 16 i + s
    const uint32_t n_iter = (a.koff + a.T + 30) / KC + 1;

    // acc: the running f32 chains; every FLUSH_S iterations they are added into the totals (244 registers with the
    // window: two waves per SIMD)
    f32x16 acc[NJT], tot[NJT];
#pragma unroll
    for (int jt = 0; jt < NJT; ++jt)
#pragma unroll
        for (int r = 0; r < 16; ++r) { acc[jt][r] = 0.0f; tot[jt][r] = 0.0f; }

    const float *hlane = a.ring + (size_t)tile * ring_tile_stride(a.R) + (size_t)(kh * 32 + cl) * 4;
    f32x4 win[SLOTS][2];
    // chunk m (relative to cb) -> window slot; the load is unconditional: past the sweep's end it fetches rows of the
    // ring that nobody uses (the ring is sized for it: ring_rows_for)
    auto load_chunk = [&](uint32_t m, f32x4 (&h)[2]) {
        uint32_t row0 = a.rb + (cb + m) * KC;              // < 2R
        row0 = row0 >= a.R ? row0 - a.R : row0;
        const f32x4 *p = (const f32x4 *)(hlane + (size_t)row0 * TILE_C);
        if constexpr (NJT == 4) {
            h[0] = __builtin_nontemporal_load(p);
            h[1] = __builtin_nontemporal_load(p + 64);
        } else {
            h[0] = p[0];
            h[1] = p[64];
        }
    };
    auto arrive = [&](uint32_t m, f32x4 (&h)[2]) {
        const uint32_t kc = (cb + m) * KC;
        if (kc + KC > a.kvalid) {                          // rows past the block's newest sample are stale
#pragma unroll
            for (int s = 0; s < KC / 2; ++s) h[s >> 2][s & 3] = kc + 8 * kh + s < a.kvalid ? h[s >> 2][s & 3] : 0.0f;
        }
        if (dirty) {
#pragma unroll
            for (int s = 0; s < KC / 2; ++s) h[s >> 2][s & 3] = finite_f32(h[s >> 2][s & 3]) ? h[s >> 2][s & 3] : 0.0f;
        }
    };
    auto wload = [&](uint32_t i, float (&w)[KC / 2]) {
#pragma unroll
        for (int s = 0; s < KC / 2; ++s) w[s] = tp[wofs + (int)(i * KC) + s];
    };
    static_for<0, SLOTS - 1>([&](auto m) { load_chunk(m.value, win[m.value]); });
    float wq[2][KC / 2];
    wload(0, wq[0]);
    static_for<0, WIN - 1>([&](auto m) { arrive(m.value, win[m.value]); });

    // iteration i, i mod SLOTS == P.  The barriers keep the order written here: consecutive MFMAs on different
    // accumulators (left alone, the scheduler groups the MFMAs of one tile, and four dependent MFMAs in a row leave the
    // pipe idle between them), the global loads and the LDS reads of the next iteration's weights in the middle of the
    // MFMA stream, a quarter / half an iteration after the waits that guard the registers they overwrite.
    auto flush = [&]() __attribute__((always_inline)) {
#pragma unroll
        for (int jt = 0; jt < NJT; ++jt)
#pragma unroll
            for (int r = 0; r < 16; ++r) { tot[jt][r] = tot[jt][r] + acc[jt][r]; acc[jt][r] = 0.0f; }
    };
    auto iter = [&](auto p_c, uint32_t i) __attribute__((always_inline)) {
        constexpr int P = decltype(p_c)::value;
        // The flush sits INSIDE the unrolled body (first iteration of every FLUSH_S / SLOTS-th body), behind a uniform
        // branch both of whose sides leave the same loads in flight.  As the end of an outer loop over segments it cost
        // 5.6 % of the sweep: at that join the paths (whole bodies, the conditional tail) have different numbers of loads
        // pending, so the compiler's counter pass made the wave wait for ALL of them -- the whole prefetch pipeline drained
        // eight times per sweep (s_waitcnt vmcnt(0); profiles/r02_fir.txt).
        if constexpr (P == 0) {
            if (i != 0 && i % FLUSH_S == 0) flush();
        }
        arrive(i + WIN - 1, win[(P + WIN - 1) % SLOTS]);              // requested D iterations ago, first used now
#pragma unroll
        for (int s = 0; s < KC / 2; ++s) {
            __builtin_amdgcn_sched_barrier(0);
            if (s == 2) load_chunk(i + SLOTS - 1, win[(P + SLOTS - 1) % SLOTS]);      // the slot chunk i - 1 has left
            if (s == 4) wload(i + 1, wq[(P + 1) & 1]);                                // (reads the table's upper pad at the end)
            if (s == 2 || s == 4) __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int jt = 0; jt < NJT; ++jt)
                acc[jt] = __builtin_amdgcn_mfma_f32_32x32x2f32(wq[P & 1][s], win[(P + 2 * jt) % SLOTS][s >> 2][s & 3], acc[jt], 0, 0, 0);
        }
        __builtin_amdgcn_sched_barrier(0);
    };
    uint32_t i = 0;
    for (; i + SLOTS <= n_iter; i += SLOTS) static_for<0, SLOTS>([&](auto p) { iter(p, i + p.value); });
    if (i < n_iter) {                                      // fewer than SLOTS left
        const uint32_t rest = n_iter - i;
        static_for<0, SLOTS - 1>([&](auto p) {
            if ((uint32_t)p.value < rest) iter(p, i + p.value);
        });
    }
    flush();
    fir_epilogue<NJT>(a, tile, c, c_ok, j0, kh, lane,
                      [&](int jt, int r) { return tot[jt][r]; });
}

// ---- steady state, split precision: f32 operands as three bf16 each, products on the bf16 matrix pipe --------------
// x = x1 + x2 + x3 and h = h1 + h2 + h3 EXACTLY (bf16 keeps f32's exponent and 8 of its 24 significant bits; each part is
// the round-to-nearest bf16 of what the parts before it left), so x h = sum of nine bf16 x bf16 products, each exact in
// f32.  The six products of order <= 2^-16 -- x1h1, x1h2, x2h1, x1h3, x3h1, x2h2 -- go through v_mfma_f32_32x32x16_bf16
// into the f32 accumulators; the three that are dropped are below 2^-24 of |x h|, the size of one f32 rounding.  One
// MFMA now covers sixteen taps where the f32 form covers two, so an output's sum sees an eighth of the f32 roundings per
// term: measured against the f64 oracle this sweep is as accurate as the f32 one (tools/fir_accuracy.py), at six bf16
// MFMAs of 32 cycles per 16 taps and tile instead of eight f32 MFMAs of 64 -- the sweep becomes HBM-bound.
// Same skewed schedule as fir_skew_kernel (one set of weights per iteration, tile jt on history chunk i + 2 jt); the
// window holds the chunks already split (3 x 4 registers each, eight slots), the f32 loads stay in flight in a ring of
// their own and are split when they arrive, between MFMA groups.  Weights: three LDS tables of bf16 PAIRS (entry m =
// parts of taps m, m + 1), so that a lane's eight consecutive taps from any start are four aligned dwords.  A workgroup
// is eight waves (two per SIMD) sharing one set of tables.
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ unsigned pack_bf16(float a, float b) {     // v_cvt_pk_bf16_f32: round to nearest even
    const bf16x2 p = __builtin_convertvector(f32x2{a, b}, bf16x2);
    unsigned u;
    __builtin_memcpy(&u, &p, 4);
    return u;
}
// eight f32 (K order) -> three vectors of eight bf16
__device__ __forceinline__ void split8(const f32x4 (&v)[2], u32x4 (&p)[3]) {
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        const float a = v[e >> 1][(e & 1) * 2], b = v[e >> 1][(e & 1) * 2 + 1];
        const unsigned p0 = pack_bf16(a, b);
        const float ra = a - __uint_as_float(p0 << 16), rb = b - __uint_as_float(p0 & 0xffff0000u);       // exact
        const unsigned p1 = pack_bf16(ra, rb);
        const float sa = ra - __uint_as_float(p1 << 16), sb = rb - __uint_as_float(p1 & 0xffff0000u);     // exact, <= 8 bits left
        p[0][e] = p0;
        p[1][e] = p1;
        p[2][e] = pack_bf16(sa, sb);
    }
}
__device__ __forceinline__ bf16x8 as_bf16x8(const u32x4 &u) {
    bf16x8 r;
    __builtin_memcpy(&r, &u, 16);
    return r;
}

constexpr int SPLIT_WAVES = 8;                 // waves per workgroup of fir_split_kernel
// LIST: the second pass of the two-part f16 sweep (tiles from its list, a persistent loop per workgroup)
template <bool LIST>
__global__ void __launch_bounds__(64 * SPLIT_WAVES) __attribute__((amdgpu_waves_per_eu(2))) fir_split_kernel(const FirMfmaArgs a) {
    constexpr int NJT = 4, WIN = 2 * (NJT - 1) + 1, SLOTS = WIN + 1, D = 4;       // SLOTS % D == 0: both rings repeat with the unrolled body
    static_assert(FLUSH % SLOTS == 0 && SLOTS % D == 0, "unroll period");
    if (a.mp.stage) fir_mixpipe_prologue(a);
    extern __shared__ unsigned tps[];          // [3][ntp4] pair tables, then the totals of tiles 1..3 per wave
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    // second pass behind the two-part f16 sweep: the tiles come from that sweep's list (usually empty: the workgroups leave
    // before they have staged anything); workgroup b takes the listed tiles 8 b .. 8 b + 7
    unsigned n_only = 0;
    if constexpr (LIST) {
        n_only = *(const volatile unsigned *)a.only_count;
        if (blockIdx.x * SPLIT_WAVES >= n_only) return;
    }
    const int ntp = (int)(PAD_LO + a.T + PAD_HI), ntp4 = (ntp + 3) & ~3;
    for (int i = tid; i < 3 * ntp4; i += 64 * SPLIT_WAVES) tps[i] = a.taps_split[i];
    __syncthreads();
  auto sweep_tile = [&](const uint32_t tile) __attribute__((always_inline)) {
    if ((size_t)tile * TILE_C >= a.N) return;
    const int cl = lane & 31, kh = lane >> 5;
    const uint32_t c = tile * TILE_C + cl;
    const bool c_ok = c < a.N;
    const bool dirty = a.nf_time[tile] > (unsigned long long)(a.t_k0 > 0 ? a.t_k0 : 0);
    const int wofs = (int)PAD_LO + 8 * kh - (int)a.koff - cl;      // table index of (iteration i, K element e) = wofs + 16 i + e
    const uint32_t n_iter = (a.koff + a.T + 30) / KC + 1;

    f32x16 acc[NJT], tot0;
    f32x4 *tl = (f32x4 *)(tps + 3 * (size_t)ntp4) + (size_t)wave * ((NJT - 1) * 4 * 64) + lane;   // [jt - 1][q][lane]
#pragma unroll
    for (int r = 0; r < 16; ++r) tot0[r] = 0.0f;
#pragma unroll
    for (int jt = 0; jt < NJT; ++jt)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[jt][r] = 0.0f;
#pragma unroll
    for (int q = 0; q < (NJT - 1) * 4; ++q) tl[q * 64] = f32x4{0.0f, 0.0f, 0.0f, 0.0f};

    const float *hlane = a.ring + (size_t)tile * ring_tile_stride(a.R) + (size_t)(kh * 32 + cl) * 4;
    auto load_chunk = [&](uint32_t m, f32x4 (&h)[2]) {     // unconditional: past the sweep's end some unused rows of the ring
        uint32_t row0 = a.rb + m * KC;                     // < 2R
        row0 = row0 >= a.R ? row0 - a.R : row0;
        const f32x4 *p = (const f32x4 *)(hlane + (size_t)row0 * TILE_C);
        h[0] = __builtin_nontemporal_load(p);
        h[1] = __builtin_nontemporal_load(p + 64);
    };
    auto arrive = [&](uint32_t m, f32x4 (&h)[2]) {
        const uint32_t kc = m * KC;
        if (kc + KC > a.kvalid) {                          // rows past the block's newest sample are stale
#pragma unroll
            for (int e = 0; e < KC / 2; ++e) h[e >> 2][e & 3] = kc + 8 * kh + e < a.kvalid ? h[e >> 2][e & 3] : 0.0f;
        }
        if (dirty) {                                       // (non-finite or huge samples: zero here, the tile is redone exactly)
#pragma unroll
            for (int e = 0; e < KC / 2; ++e) h[e >> 2][e & 3] = finite_f32(h[e >> 2][e & 3]) ? h[e >> 2][e & 3] : 0.0f;
        }
    };
    auto wload = [&](uint32_t i, u32x4 (&w)[3]) {
#pragma unroll
        for (int sp = 0; sp < 3; ++sp)
#pragma unroll
            for (int e = 0; e < 4; ++e) w[sp][e] = tps[sp * ntp4 + wofs + (int)(i * KC) + 2 * e];
    };
    u32x4 win[SLOTS][3];
    f32x4 fly[D][2];
    {   // chunks 0 .. WIN-1 split into the window, WIN .. WIN+D-1 in flight
        f32x4 t[WIN][2];
        static_for<0, WIN>([&](auto m) { load_chunk(m.value, t[m.value]); });
        static_for<0, D>([&](auto m) { load_chunk(WIN + m.value, fly[(WIN + m.value) % D]); });
        static_for<0, WIN>([&](auto m) {
            arrive(m.value, t[m.value]);
            split8(t[m.value], win[m.value]);
        });
    }
    u32x4 wq[2][3];
    wload(0, wq[0]);

    // iteration i, i mod SLOTS == P: six groups of NJT MFMAs (one per product term), the other work pinned between them
    auto flush = [&]() __attribute__((always_inline)) {
#pragma unroll
        for (int r = 0; r < 16; ++r) { tot0[r] = tot0[r] + acc[0][r]; acc[0][r] = 0.0f; }
#pragma unroll
        for (int jt = 1; jt < NJT; ++jt)
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                f32x4 t = tl[((jt - 1) * 4 + q) * 64];
#pragma unroll
                for (int e = 0; e < 4; ++e) { t[e] = t[e] + acc[jt][q * 4 + e]; acc[jt][q * 4 + e] = 0.0f; }
                tl[((jt - 1) * 4 + q) * 64] = t;
            }
    };
    auto iter = [&](auto p_c, uint32_t i) __attribute__((always_inline)) {
        constexpr int P = decltype(p_c)::value;
        if constexpr (P == 0) {                            // the flush inside the unrolled body: see fir_skew_kernel
            if (i != 0 && i % FLUSH == 0) flush();
        }
        constexpr int TX[6] = {0, 0, 1, 0, 2, 1}, TH[6] = {0, 1, 0, 2, 0, 1};      // x part, h part of the six terms
        f32x4 (&in)[2] = fly[(P + WIN) % D];                                       // chunk i + WIN: lands in window slot (P + WIN) % SLOTS
#pragma unroll
        for (int t = 0; t < 6; ++t) {
            __builtin_amdgcn_sched_barrier(0);
            if (t == 0) arrive(i + WIN, in);
            if (t == 1) split8(in, win[(P + WIN) % SLOTS]);
            if (t == 3) load_chunk(i + WIN + D, in);
            if (t == 4) wload(i + 1, wq[(P + 1) & 1]);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int jt = 0; jt < NJT; ++jt)
                acc[jt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(as_bf16x8(wq[P & 1][TH[t]]), as_bf16x8(win[(P + 2 * jt) % SLOTS][TX[t]]),
                                                                  acc[jt], 0, 0, 0);
        }
        __builtin_amdgcn_sched_barrier(0);
    };
    uint32_t i = 0;
    for (; i + SLOTS <= n_iter; i += SLOTS) static_for<0, SLOTS>([&](auto p) { iter(p, i + p.value); });
    if (i < n_iter) {                                      // fewer than SLOTS left
        const uint32_t rest = n_iter - i;
        static_for<0, SLOTS - 1>([&](auto p) {
            if ((uint32_t)p.value < rest) iter(p, i + p.value);
        });
    }
    flush();
    fir_epilogue<NJT>(a, tile, c, c_ok, 0, kh, lane,
                      [&](int jt, int r) { return jt == 0 ? tot0[r] : tl[((jt > 0 ? jt - 1 : 0) * 4 + (r >> 2)) * 64][r & 3]; });
  };
  // (LIST as a persistent loop over the list -- a launch of 256 workgroups instead of one per eight tiles -- made the compiler
  // spill 136 registers in the sweep; one workgroup per eight LISTED tiles it is, the others leave at once)
  if constexpr (LIST) {
      if (blockIdx.x * SPLIT_WAVES + wave < n_only) sweep_tile(a.only_tiles[blockIdx.x * SPLIT_WAVES + wave]);
  } else {
      sweep_tile(blockIdx.x * SPLIT_WAVES + wave);
  }
}


// ---- steady state, two-part split: f32 operands as f16 hi + f16 lo, THREE products per term ----------------------------
// The bf16 x 3 sweep above is bound by the matrix pipe at the socket's power limit (71 % busy at 1.63 GHz: six MFMAs per 16
// taps and tile).  f16 keeps 11 significant bits where bf16 keeps 8, so TWO parts hold 22: with x' = x 2^14 and h' = h 2^p
// (p: the largest tap lands in [2^14, 2^15)),  x' = xh + xl,  h' = hh + hl  (each the round-to-nearest f16 of what is left),
//     x' h' = xh hh + xh hl + xl hh   (+ xl hl, below 2^-22 of the product: dropped)
// -- three v_mfma_f32_32x32x16_f16 per 16 taps and tile instead of six, each product exact in f32, the sum scaled back by the
// exact power of two.  Representation error 8e-8 relative RMS at 4096 taps (numpy model, tools/fir_accuracy.py); with the f32
// accumulation the sweep measures the same ~3e-7 against the f64 oracle as the other two.  f16 has five exponent bits where
// bf16 has f32's eight, which is why this needs a RANGE: samples up to 3.998 in magnitude (x' <= 65504), and an absolute floor
// of 2^-39 below which nothing is resolved (the smallest f16 subnormal / 2^14; the MFMA honours f16 subnormals: measured,
// tools/micro/mfma_f16_denorm.hip).  Every wave therefore tracks the peak |x| of each of its channels over the window it
// actually swept; a tile with a channel whose peak is >= 3.998 or (non-zero and) < 2^-13 is LISTED, and the bf16 x 3 sweep --
// launched right behind, over the listed tiles only, usually none -- redoes it.  Audio at ordinary levels never leaves the
// fast path; integers beyond 3, fades into the noise floor and denormal dust take the slower sweep: same bar either way.
// Schedule, window, weights-in-LDS and epilogue are fir_split_kernel's with two parts instead of three.
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));
#ifndef DSPFX_HALF_D
#define DSPFX_HALF_D 4
#endif
constexpr int HALF_D = DSPFX_HALF_D;          // history chunks in flight per wave behind the window (A/B builds: 4 or 8)
constexpr float HALF_X_SCALE = 0x1p14f, HALF_PEAK_MAX = 3.998046875f * 0x1p14f, HALF_PEAK_MIN = 0x1p-13f * 0x1p14f;   // in units of x'

__device__ __forceinline__ unsigned pack_f16(float a, float b) {      // v_cvt_pk_f16_f32: round to nearest even (measured)
    const f16x2 p = __builtin_convertvector(f32x2{a, b}, f16x2);
    unsigned u;
    __builtin_memcpy(&u, &p, 4);
    return u;
}
__device__ __forceinline__ f32x2 unpack_f16(unsigned u) {
    f16x2 p;
    __builtin_memcpy(&p, &u, 4);
    return __builtin_convertvector(p, f32x2);
}
// four f32 (one 16-byte piece, K order) -> two pairs of dwords (hi, lo of x 2^14); the lane's running peak |x 2^14|.
// lo = x' - hi is formed by ONE mixed-precision fma per element (v_fma_mix_f32: the f16 operand is widened inside the
// instruction) instead of a conversion and a subtraction; it is exact either way (<= 13 significant bits).
// (`half` = which piece of the chunk: elements 2 half, 2 half + 1 of the two part vectors)
// lo = x' - hi: v_fma_mix_f32 d, hi.{lo,hi half as f16}, -1.0, x' -- the f16 operand is widened inside the instruction; spelled
// in assembly because the compiler turns fma(ext(h), -1, a) back into a conversion and a (packed) subtraction.
__device__ __forceinline__ float sub_f16_lo(float a, unsigned h) {
    float r;
    asm("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel:[0,0,0] op_sel_hi:[1,0,0]" : "=v"(r) : "v"(h), "v"(a));
    return r;
}
__device__ __forceinline__ float sub_f16_hi(float a, unsigned h) {
    float r;
    asm("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "=v"(r) : "v"(h), "v"(a));
    return r;
}
__device__ __forceinline__ void split4h(const f32x4 &v, u32x4 (&p)[2], const int half, float &peak) {
#ifdef DSPFX_HALF_NOSPLIT      // experiment build ONLY (tools/r05_fir_nosplit_ab.sh): what would the sweep cost if the history arrived already split
    // into packed f16 parts (a second ring written by the append pass)?  The f32 bits are fed to the matrix pipe as they are --
    // the results are garbage, the instruction stream is the packed-ring sweep's: an upper bound on what that design can gain.
    p[0][2 * half] = __float_as_uint(v[0]);
    p[0][2 * half + 1] = __float_as_uint(v[1]);
    p[1][2 * half] = __float_as_uint(v[2]);
    p[1][2 * half + 1] = __float_as_uint(v[3]);
    return;
#endif
    const float a = v[0] * HALF_X_SCALE, b = v[1] * HALF_X_SCALE, c = v[2] * HALF_X_SCALE, d = v[3] * HALF_X_SCALE;
    const unsigned h0 = pack_f16(a, b), h1 = pack_f16(c, d);
    const unsigned l0 = pack_f16(sub_f16_lo(a, h0), sub_f16_hi(b, h0)), l1 = pack_f16(sub_f16_lo(c, h1), sub_f16_hi(d, h1));
    peak = __builtin_fmaxf(__builtin_fmaxf(peak, __builtin_fmaxf(__builtin_fabsf(a), __builtin_fabsf(b))), __builtin_fmaxf(__builtin_fabsf(c), __builtin_fabsf(d)));
    p[0][2 * half] = h0;
    p[0][2 * half + 1] = h1;
    p[1][2 * half] = l0;
    p[1][2 * half + 1] = l1;
}
// eight f32 (K order) -> two vectors of eight f16
__device__ __forceinline__ void split8h(const f32x4 (&v)[2], u32x4 (&p)[2], float &peak) {
    split4h(v[0], p, 0, peak);
    split4h(v[1], p, 1, peak);
}
__device__ __forceinline__ f16x8 as_f16x8(const u32x4 &u) {
    f16x8 r;
    __builtin_memcpy(&r, &u, 16);
    return r;
}

__global__ void __launch_bounds__(64 * SPLIT_WAVES) __attribute__((amdgpu_waves_per_eu(2))) fir_half_kernel(const FirMfmaArgs a) {
    constexpr int NJT = 4, WIN = 2 * (NJT - 1) + 1, SLOTS = WIN + 1, D = HALF_D;       // SLOTS % D == 0: both rings repeat with the unrolled body
    static_assert(FLUSH % SLOTS == 0 && SLOTS % D == 0, "unroll period");
    if (a.mp.stage) fir_mixpipe_prologue(a);
    if (blockIdx.x == 0 && threadIdx.x == 0) *a.redo_clear = 0u;                  // the next block's list starts empty
    extern __shared__ unsigned tps[];          // [2][ntp4] pair tables, then the totals of tiles 1..3 per wave
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int ntp = (int)(PAD_LO + a.T + PAD_HI), ntp4 = (ntp + 3) & ~3;
    for (int i = tid; i < 2 * ntp4; i += 64 * SPLIT_WAVES) tps[i] = a.taps_half[i];
    __syncthreads();
    const uint32_t tile = blockIdx.x * SPLIT_WAVES + wave;
    if ((size_t)tile * TILE_C >= a.N) return;
    const int cl = lane & 31, kh = lane >> 5;
    const uint32_t c = tile * TILE_C + cl;
    const bool c_ok = c < a.N;
    const bool dirty = a.nf_time[tile] > (unsigned long long)(a.t_k0 > 0 ? a.t_k0 : 0);
    const int wofs = (int)PAD_LO + 8 * kh - (int)a.koff - cl;      // table index of (iteration i, K element e) = wofs + 16 i + e
    const uint32_t n_iter = (a.koff + a.T + 30) / KC + 1;

    f32x16 acc[NJT], tot0;
    f32x4 *tl = (f32x4 *)(tps + 2 * (size_t)ntp4) + (size_t)wave * ((NJT - 1) * 4 * 64) + lane;   // [jt - 1][q][lane]
#pragma unroll
    for (int r = 0; r < 16; ++r) tot0[r] = 0.0f;
#pragma unroll
    for (int jt = 0; jt < NJT; ++jt)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[jt][r] = 0.0f;
#pragma unroll
    for (int q = 0; q < (NJT - 1) * 4; ++q) tl[q * 64] = f32x4{0.0f, 0.0f, 0.0f, 0.0f};

    const float *hlane = a.ring + (size_t)tile * ring_tile_stride(a.R) + (size_t)(kh * 32 + cl) * 4;
    auto load_chunk = [&](uint32_t m, f32x4 (&h)[2]) {     // unconditional: past the sweep's end some unused rows of the ring
        uint32_t row0 = a.rb + m * KC;                     // < 2R
        row0 = row0 >= a.R ? row0 - a.R : row0;
        const f32x4 *p = (const f32x4 *)(hlane + (size_t)row0 * TILE_C);
        h[0] = __builtin_nontemporal_load(p);
        h[1] = __builtin_nontemporal_load(p + 64);
    };
    auto arrive = [&](uint32_t m, f32x4 (&h)[2]) {
        const uint32_t kc = m * KC;
        if (kc + KC > a.kvalid) {                          // rows past the block's newest sample are stale
#pragma unroll
            for (int e = 0; e < KC / 2; ++e) h[e >> 2][e & 3] = kc + 8 * kh + e < a.kvalid ? h[e >> 2][e & 3] : 0.0f;
        }
        if (dirty) {                                       // (non-finite or huge samples: zero here, the tile is redone exactly)
#pragma unroll
            for (int e = 0; e < KC / 2; ++e) h[e >> 2][e & 3] = finite_f32(h[e >> 2][e & 3]) ? h[e >> 2][e & 3] : 0.0f;
        }
    };
    auto wload = [&](uint32_t i, u32x4 (&w)[2]) {
#pragma unroll
        for (int sp = 0; sp < 2; ++sp)
#pragma unroll
            for (int e = 0; e < 4; ++e) w[sp][e] = tps[sp * ntp4 + wofs + (int)(i * KC) + 2 * e];
    };
    u32x4 win[SLOTS][2];
    f32x4 fly[D][2];
    float peak = 0.0f;
    {   // chunks 0 .. WIN-1 split into the window, WIN .. WIN+D-1 in flight
        f32x4 t[WIN][2];
        static_for<0, WIN>([&](auto m) { load_chunk(m.value, t[m.value]); });
        static_for<0, D>([&](auto m) { load_chunk(WIN + m.value, fly[(WIN + m.value) % D]); });
        static_for<0, WIN>([&](auto m) {
            arrive(m.value, t[m.value]);
            split8h(t[m.value], win[m.value], peak);
        });
    }
    u32x4 wq[2][2];
    wload(0, wq[0]);

    auto flush = [&]() __attribute__((always_inline)) {
#pragma unroll
        for (int r = 0; r < 16; ++r) { tot0[r] = tot0[r] + acc[0][r]; acc[0][r] = 0.0f; }
#pragma unroll
        for (int jt = 1; jt < NJT; ++jt)
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                f32x4 t = tl[((jt - 1) * 4 + q) * 64];
#pragma unroll
                for (int e = 0; e < 4; ++e) { t[e] = t[e] + acc[jt][q * 4 + e]; acc[jt][q * 4 + e] = 0.0f; }
                tl[((jt - 1) * 4 + q) * 64] = t;
            }
    };
    // iteration i, i mod SLOTS == P: three groups of NJT MFMAs (one per product term), the other work pinned between them
    auto iter = [&](auto p_c, uint32_t i) __attribute__((always_inline)) {
        constexpr int P = decltype(p_c)::value;
        if constexpr (P == 0) {                            // the flush inside the unrolled body: see fir_skew_kernel
            if (i != 0 && i % FLUSH == 0) flush();
        }
        constexpr int TX[3] = {0, 0, 1}, TH[3] = {0, 1, 0};                        // x part, h part of the three terms
        f32x4 (&in)[2] = fly[(P + WIN) % D];                                       // chunk i + WIN: lands in window slot (P + WIN) % SLOTS
#pragma unroll
        for (int t = 0; t < 3; ++t) {
            __builtin_amdgcn_sched_barrier(0);
            // (the chunk split here is first used by the NEXT iteration: its work can sit anywhere among this one's groups)
            u32x4 (&nw)[2] = win[(P + WIN) % SLOTS];
            if (t == 0) {
                arrive(i + WIN, in);
                split4h(in[0], nw, 0, peak);
            }
            if (t == 1) {
                asm volatile("" : "+v"(in[1][0]), "+v"(in[1][1]), "+v"(in[1][2]), "+v"(in[1][3]));      // (keeps this half's split HERE: the compiler merges it into the first otherwise)
                split4h(in[1], nw, 1, peak);
                load_chunk(i + WIN + D, in);
            }
            if (t == 2) wload(i + 1, wq[(P + 1) & 1]);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int jt = 0; jt < NJT; ++jt)
                acc[jt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(as_f16x8(wq[P & 1][TH[t]]), as_f16x8(win[(P + 2 * jt) % SLOTS][TX[t]]),
                                                                 acc[jt], 0, 0, 0);
        }
        __builtin_amdgcn_sched_barrier(0);
    };
    uint32_t i = 0;
    for (; i + SLOTS <= n_iter; i += SLOTS) static_for<0, SLOTS>([&](auto p) { iter(p, i + p.value); });
    if (i < n_iter) {                                      // fewer than SLOTS left
        const uint32_t rest = n_iter - i;
        static_for<0, SLOTS - 1>([&](auto p) {
            if ((uint32_t)p.value < rest) iter(p, i + p.value);
        });
    }
    flush();
    // Could this sweep serve every channel of the tile?  (The peak covers the rows the window really held, a few chunks past
    // the sweep's end included: conservative.)  NaN compares false: such a tile is listed too (and redone exactly besides).
    peak = __builtin_fmaxf(peak, __shfl_xor(peak, 32));                            // the channel's two lanes
    const bool ok = peak == 0.0f || (peak >= HALF_PEAK_MIN && peak <= HALF_PEAK_MAX);
    if (__ballot(ok) != ~0ull && lane == 0) a.redo_tiles[atomicAdd(a.redo_count, 1u)] = tile;
    const float us = a.half_unscale;
    fir_epilogue<NJT>(a, tile, c, c_ok, 0, kh, lane,
                      [&](int jt, int r) { return (jt == 0 ? tot0[r] : tl[((jt > 0 ? jt - 1 : 0) * 4 + (r >> 2)) * 64][r & 3]) * us; });
}

// ---- the same sweep over the PACKED history (round 5) -------------------------------------------------------------------------
// The history arrives already split (FirState::ringh, written by fir_append2_kernel): a chunk's two 16-byte loads per lane ARE the two
// B operands -- no conversion, no subtraction, no peak scan in the loop (fir_half_kernel does ~30 vector-ALU instructions per lane
// and chunk for them, 33 times per sample at 4096 taps, inside a power-limited kernel); the window's peak per channel comes from
// the append pass' epoch table.  With nothing to do to a chunk when it arrives, the window and the chunks in flight share ONE ring
// of NS register slots that the loads write directly: chunk m lives in slot m % NS from its load (iteration m - NS + 1) to its
// last use (iteration m), the unrolled body has NS iterations, and D = NS - 8 chunks are in flight behind the seven live ones.
template <int NS>
__global__ void __launch_bounds__(64 * SPLIT_WAVES) __attribute__((amdgpu_waves_per_eu(2))) fir_halfp_kernel(const FirMfmaArgs a) {
    constexpr int NJT = 4, WIN = 2 * (NJT - 1) + 1, D = NS - WIN - 1;
    constexpr int HF = (FLUSH + NS - 1) / NS * NS;          // chunks per accumulator flush: a multiple of the unroll period (36 / 32)
    static_assert(D >= 2 && NS % 2 == 0, "ring of chunk slots");
    if (a.mp.stage) fir_mixpipe_prologue(a);
    if (blockIdx.x == 0 && threadIdx.x == 0) *a.redo_clear = 0u;                  // the next block's list starts empty
    extern __shared__ unsigned tps[];          // [2][ntp4] pair tables, then the totals of tiles 1..3 per wave
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int ntp = (int)(PAD_LO + a.T + PAD_HI), ntp4 = (ntp + 3) & ~3;
    for (int i = tid; i < 2 * ntp4; i += 64 * SPLIT_WAVES) tps[i] = a.taps_half[i];
    __syncthreads();
    const uint32_t tile = blockIdx.x * SPLIT_WAVES + wave;
    if ((size_t)tile * TILE_C >= a.N) return;
    const int cl = lane & 31, kh = lane >> 5;
    const uint32_t c = tile * TILE_C + cl;
    const bool c_ok = c < a.N;
    const int wofs = (int)PAD_LO + 8 * kh - (int)a.koff - cl;      // table index of (iteration i, K element e) = wofs + 16 i + e
    const uint32_t n_iter = (a.koff + a.T + 30) / KC + 1;

    // Could this sweep serve every channel of the tile?  Decided HERE, before the sweep, from the append pass' per-epoch peaks of the
    // epochs that hold the window's sample times [max(t_k0, 0), t_k0 + kvalid) -- at most 34 entries of 128 bytes per tile, the
    // channel's two lanes take every other one -- and kept as one wave-uniform mask (no register lives through the sweep).
    //   `peak`: over every epoch that overlaps the window.  An upper bound of the window's true peak (a boundary epoch reaches up
    //           to 127 samples past either end of it): the right side for HALF_PEAK_MAX, for non-finite samples (infinite peak)
    //           and for "all zero".
    //   `pin` : over the epochs that lie INSIDE the window -- a LOWER bound of the true peak, the right side for HALF_PEAK_MIN.
    //           Only when the sole loud samples sit in the boundary epochs (a loud-to-quiet transition, the first blocks of a
    //           stream) is their in-window part scanned exactly in the f32 ring (with `peak` on both sides a loud
    //           sample just outside the window hid a window below 2^-13, whose f16 lo parts go subnormal; fir_half_kernel,
    //           which measures the window it sweeps, would have redone it).
    bool ok = true;
    {
        float peak = 0.0f, pin = 0.0f;
        const long long t_last = a.t_k0 + (long long)a.kvalid - 1;
        if (t_last >= 0) {
            const unsigned long long t_lo = (unsigned long long)(a.t_k0 > 0 ? a.t_k0 : 0);
            const unsigned long long e_lo = t_lo / EPOCH, e_hi = (unsigned long long)t_last / EPOCH;
            const float *pk = a.peaks + (size_t)tile * a.peak_slots * TILE_C + cl;
            for (unsigned long long e = e_lo + kh; e <= e_hi; e += 2) {
                const float v = pk[(size_t)(e % a.peak_slots) * TILE_C];
                peak = __builtin_fmaxf(peak, v);
                if (e * EPOCH >= t_lo && (e + 1) * EPOCH - 1 <= (unsigned long long)t_last) pin = __builtin_fmaxf(pin, v);
            }
            peak = __builtin_fmaxf(peak, __shfl_xor(peak, 32));                    // the channel's two lanes
            pin = __builtin_fmaxf(pin, __shfl_xor(pin, 32));
            if (c_ok && pin < HALF_PEAK_MIN && peak >= HALF_PEAK_MIN && peak <= HALF_PEAK_MAX) {
                // the boundary epochs' samples inside the window, exactly (<= 2 x 127 per channel, split between its two lanes)
                const float *rc = a.ring + (size_t)tile * ring_tile_stride(a.R);
                auto scan = [&](unsigned long long e) {
                    const unsigned long long b0 = e * EPOCH > t_lo ? e * EPOCH : t_lo;
                    const unsigned long long b1 = (e + 1) * EPOCH - 1 < (unsigned long long)t_last ? (e + 1) * EPOCH - 1 : (unsigned long long)t_last;
                    for (unsigned long long t = b0 + kh; t <= b1; t += 2)
                        pin = __builtin_fmaxf(pin, __builtin_fabsf(rc[ring_in_tile((uint32_t)(t % a.R), cl)]) * HALF_APPEND_SCALE);
                };
                if (e_lo * EPOCH < t_lo || (e_lo + 1) * EPOCH - 1 > (unsigned long long)t_last) scan(e_lo);
                if (e_hi != e_lo && (e_hi + 1) * EPOCH - 1 > (unsigned long long)t_last) scan(e_hi);
            }
            pin = __builtin_fmaxf(pin, __shfl_xor(pin, 32));
        }
        ok = peak == 0.0f || (pin >= HALF_PEAK_MIN && peak <= HALF_PEAK_MAX);
    }
    const bool tile_ok = __ballot(ok) == ~0ull;

    f32x16 acc[NJT], tot0;
    f32x4 *tl = (f32x4 *)(tps + 2 * (size_t)ntp4) + (size_t)wave * ((NJT - 1) * 4 * 64) + lane;   // [jt - 1][q][lane]
#pragma unroll
    for (int r = 0; r < 16; ++r) tot0[r] = 0.0f;
#pragma unroll
    for (int jt = 0; jt < NJT; ++jt)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[jt][r] = 0.0f;
#pragma unroll
    for (int q = 0; q < (NJT - 1) * 4; ++q) tl[q * 64] = f32x4{0.0f, 0.0f, 0.0f, 0.0f};

    // (the two rings share tile stride and chunk geometry: the lane's offset inside a chunk is the same in both)
    const unsigned *hlane = a.ringh + (size_t)tile * ring_tile_stride(a.R) + (size_t)(kh * 32 + cl) * 4;
    auto load_chunk = [&](uint32_t m, u32x4 (&h)[2]) {     // unconditional: past the sweep's end some unused rows of the ring
        uint32_t row0 = a.rb + m * KC;                     // < 2R
        row0 = row0 >= a.R ? row0 - a.R : row0;
        const u32x4 *p = (const u32x4 *)(hlane + (size_t)row0 * TILE_C);
        h[0] = __builtin_nontemporal_load(p);              // the hi parts of the lane's eight rows
        h[1] = __builtin_nontemporal_load(p + 64);         // the lo parts
    };
    auto mask_tail = [&](uint32_t m, u32x4 (&h)[2]) {      // rows past the block's newest sample are stale: dword d holds K elements 2 d, 2 d + 1
        const uint32_t kc = m * KC;
        if (kc + KC > a.kvalid) {
#pragma unroll
            for (int d = 0; d < 4; ++d) {
                const unsigned keep = (kc + 8 * kh + 2 * d < a.kvalid ? 0x0000ffffu : 0u) | (kc + 8 * kh + 2 * d + 1 < a.kvalid ? 0xffff0000u : 0u);
                h[0][d] &= keep;
                h[1][d] &= keep;
            }
        }
    };                                                      // (non-finite samples were written as zeros by the append pass)
    auto wload = [&](uint32_t i, u32x4 (&w)[2]) {
#pragma unroll
        for (int sp = 0; sp < 2; ++sp)
#pragma unroll
            for (int e = 0; e < 4; ++e) w[sp][e] = tps[sp * ntp4 + wofs + (int)(i * KC) + 2 * e];
    };
    u32x4 buf[NS][2];
    static_for<0, NS - 1>([&](auto m) { load_chunk(m.value, buf[m.value]); });          // chunks 0 .. WIN + D - 1
    static_for<0, WIN>([&](auto m) { mask_tail(m.value, buf[m.value]); });
    u32x4 wq[2][2];
    wload(0, wq[0]);

    auto flush = [&]() __attribute__((always_inline)) {
#pragma unroll
        for (int r = 0; r < 16; ++r) { tot0[r] = tot0[r] + acc[0][r]; acc[0][r] = 0.0f; }
#pragma unroll
        for (int jt = 1; jt < NJT; ++jt)
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                f32x4 t = tl[((jt - 1) * 4 + q) * 64];
#pragma unroll
                for (int e = 0; e < 4; ++e) { t[e] = t[e] + acc[jt][q * 4 + e]; acc[jt][q * 4 + e] = 0.0f; }
                tl[((jt - 1) * 4 + q) * 64] = t;
            }
    };
    // iteration i, i mod NS == P: three groups of NJT MFMAs (one per product term) on the chunks i, i + 2, i + 4, i + 6
    auto iter = [&](auto p_c, uint32_t i) __attribute__((always_inline)) {
        constexpr int P = decltype(p_c)::value;
        if constexpr (P == 0) {                            // the flush inside the unrolled body: see fir_skew_kernel
            if (i != 0 && i % HF == 0) flush();
        }
        constexpr int TX[3] = {0, 0, 1}, TH[3] = {0, 1, 0};                        // x part, h part of the three terms
#pragma unroll
        for (int t = 0; t < 3; ++t) {
            __builtin_amdgcn_sched_barrier(0);
            if (t == 0) mask_tail(i + WIN, buf[(P + WIN) % NS]);                   // chunk i + WIN is first used by the next iteration
            if (t == 1) load_chunk(i + NS - 1, buf[(P + NS - 1) % NS]);            // into the slot chunk i - 1 has just left
            if (t == 2) wload(i + 1, wq[(P + 1) & 1]);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int jt = 0; jt < NJT; ++jt)
                acc[jt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(as_f16x8(wq[P & 1][TH[t]]), as_f16x8(buf[(P + 2 * jt) % NS][TX[t]]), acc[jt], 0, 0, 0);
        }
        __builtin_amdgcn_sched_barrier(0);
    };
    uint32_t i = 0;
    for (; i + NS <= n_iter; i += NS) static_for<0, NS>([&](auto p) { iter(p, i + p.value); });
    if (i < n_iter) {                                      // fewer than NS left
        const uint32_t rest = n_iter - i;
        static_for<0, NS - 1>([&](auto p) {
            if ((uint32_t)p.value < rest) iter(p, i + p.value);
        });
    }
    flush();
    // a tile this sweep could not serve (decided above) is listed for the bf16 x 3 second pass; a non-finite sample made its
    // epoch's peak infinite: listed too (and redone exactly besides)
    if (!tile_ok && lane == 0) a.redo_tiles[atomicAdd(a.redo_count, 1u)] = tile;
    const float us = a.half_unscale;
    fir_epilogue<NJT>(a, tile, c, c_ok, 0, kh, lane,
                      [&](int jt, int r) { return (jt == 0 ? tot0[r] : tl[((jt > 0 ? jt - 1 : 0) * 4 + (r >> 2)) * 64][r & 3]) * us; });
}

// old ring -> new ring for the sample times [t_begin, t_end): a tap reload that needs more rows
__global__ void __launch_bounds__(256) fir_rebase_kernel(const float *src, float *dst, uint32_t tiles, uint32_t R_src, uint32_t R_dst,
                                                         unsigned long long t_begin, unsigned long long t_end) {
    const size_t per_t = (size_t)tiles * TILE_C;
    const size_t total = (size_t)(t_end - t_begin) * per_t;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
        const unsigned long long t = t_begin + i / per_t;
        const uint32_t c = (uint32_t)(i % per_t);
        dst[ring_at(c, (uint32_t)(t % R_dst), R_dst)] = src[ring_at(c, (uint32_t)(t % R_src), R_src)];
    }
}

// dense[k][c] <-> ring row of time t0 + k (state export / import); times before 0 read as zeros
__global__ void __launch_bounds__(256) fir_rows_kernel(float *ring, float *dense, uint32_t N, uint32_t R, long long t0, uint32_t nrows,
                                                       int to_dense) {
    const uint32_t c = blockIdx.x * 256 + threadIdx.x;
    if (c >= N) return;
    for (uint32_t k = blockIdx.y; k < nrows; k += gridDim.y) {
        const long long t = t0 + k;
        float *cell = ring + ring_at(c, (uint32_t)((t % (long long)R + (long long)R) % (long long)R), R);
        if (to_dense) dense[(size_t)k * N + c] = t < 0 ? 0.0f : *cell;
        else if (t >= 0) *cell = dense[(size_t)k * N + c];
    }
}

// nf_time[tile] = 1 + time of the newest non-finite sample among the times [0, held) (an imported history)
__global__ void __launch_bounds__(256) fir_flag_scan_kernel(const float *ring, unsigned long long *nf_time, uint32_t N, uint32_t R,
                                                            unsigned long long held) {
    const uint32_t tile = blockIdx.x;
    unsigned long long best = 0;
    for (unsigned long long i = threadIdx.x; i < held * TILE_C; i += 256) {
        const unsigned long long t = i / TILE_C;
        const uint32_t c = tile * TILE_C + (uint32_t)(i % TILE_C);
        if (c < N && !finite_f32(ring[ring_at(c, (uint32_t)(t % R), R)])) best = t + 1 > best ? t + 1 : best;
    }
    if (best) atomicMax(&nf_time[tile], best);
}

static uint32_t ring_rows_for(uint64_t held, uint32_t n_taps, uint32_t max_frames) {
    // rows the sweep may touch: the deque (held samples, at least T-1), the block, the alignment pad and one chunk of slack
    // (the skewed sweep prefetches up to eight chunks past the newest row of a full slice)
    uint64_t need = std::max<uint64_t>(held + 1, n_taps) + std::max<uint32_t>(max_frames, SLICE) + 24 * KC;
    if (need < 4 * KC) need = 4 * KC;
    // whole 16-row groups (ring layout above); the tile stride (R + 1) * 128 B is then an odd multiple of 128 B (measured
    // on the row-major ring: a stride that is a multiple of 4 KiB ran the 4096-tap sweep 3.4 % slower)
    return (uint32_t)((need + KC - 1) / KC * KC);
}

constexpr size_t LDS_PER_CU = 160 * 1024;
// round-to-nearest-even bf16 of a finite f32, as its 16 bits
static uint32_t bf16_bits(float v) {
    uint32_t u;
    memcpy(&u, &v, 4);
    return (u + 0x7fffu + ((u >> 16) & 1u)) >> 16;
}
static float bf16_value(uint32_t b) {
    const uint32_t u = b << 16;
    float v;
    memcpy(&v, &u, 4);
    return v;
}
static size_t split_lds_bytes(uint32_t n_taps) {
    const size_t ntp4 = ((size_t)PAD_LO + n_taps + PAD_HI + 3) & ~(size_t)3;
    return 3 * ntp4 * sizeof(unsigned) + (size_t)SPLIT_WAVES * 3 * 4 * 64 * sizeof(f32x4);
}
static size_t half_lds_bytes(uint32_t n_taps) {
    const size_t ntp4 = ((size_t)PAD_LO + n_taps + PAD_HI + 3) & ~(size_t)3;
    return 2 * ntp4 * sizeof(unsigned) + (size_t)SPLIT_WAVES * 3 * 4 * 64 * sizeof(f32x4);
}
static size_t tap_table_bytes(uint32_t n_taps) { return ((size_t)PAD_LO + n_taps + PAD_HI) * sizeof(float); }
static size_t skew_lds_bytes(uint32_t n_taps, int) { return tap_table_bytes(n_taps); }

static void free_packed(FirState &s) {
    if (s.ringh) (void)hipFree(s.ringh);
    if (s.peaks) (void)hipFree(s.peaks);
    s.ringh = nullptr;
    s.peaks = nullptr;
    s.peak_slots = 0;
    s.packed_ok = false;
}
// the packed history + peak table for a ring of s.R rows (zeroed: consistent with an empty or all-zero f32 ring only)
static int alloc_packed(FirState &s) {
    free_packed(s);
    if (s.env.packed == 0) return 0;
    const size_t bytes = ring_bytes_for(s.tiles, s.R);
    s.peak_slots = s.R / EPOCH + 3;
    FIRCHK(hipMalloc((void **)&s.ringh, bytes));
    FIRCHK(hipMemset(s.ringh, 0, bytes));
    FIRCHK(hipMalloc((void **)&s.peaks, (size_t)s.tiles * s.peak_slots * TILE_C * sizeof(float)));
    FIRCHK(hipMemset(s.peaks, 0, (size_t)s.tiles * s.peak_slots * TILE_C * sizeof(float)));
    return 0;
}

static int upload_taps(FirState &s, const double *taps_reversed, uint32_t n_taps) {
    if (s.taps64) (void)hipFree(s.taps64);
    if (s.taps32) (void)hipFree(s.taps32);
    if (s.taps_split) (void)hipFree(s.taps_split);
    if (s.taps_half) (void)hipFree(s.taps_half);
    s.taps64 = nullptr;
    s.taps32 = nullptr;
    s.taps_split = nullptr;
    s.taps_half = nullptr;
    s.T = n_taps;
    s.pad_lo = PAD_LO;
    s.pad_hi = PAD_HI;
    FIRCHK(hipMalloc((void **)&s.taps64, (size_t)n_taps * sizeof(double)));
    FIRCHK(hipMemcpy(s.taps64, taps_reversed, (size_t)n_taps * sizeof(double), hipMemcpyHostToDevice));
    std::vector<float> t32((size_t)PAD_LO + n_taps + PAD_HI, 0.0f);
    for (uint32_t i = 0; i < n_taps; ++i) t32[PAD_LO + i] = (float)taps_reversed[i];
    FIRCHK(hipMalloc((void **)&s.taps32, t32.size() * sizeof(float)));
    FIRCHK(hipMemcpy(s.taps32, t32.data(), t32.size() * sizeof(float), hipMemcpyHostToDevice));
    // split-precision tables: each padded tap as three bf16 parts (exact: tap == p0 + p1 + p2), stored as pairs (m, m + 1)
    bool tame = true;
    for (float t : t32) tame = tame && std::fabs(t) < 0x1p127f;
    if (tame && split_lds_bytes(n_taps) <= LDS_PER_CU) {
        const size_t ntp = t32.size(), ntp4 = (ntp + 3) & ~(size_t)3;
        std::vector<uint32_t> part[3];
        for (auto &v : part) v.assign(ntp + 1, 0);
        for (size_t m = 0; m < ntp; ++m) {
            float r = t32[m];
            for (int k = 0; k < 3; ++k) {
                part[k][m] = bf16_bits(r);
                r -= bf16_value(part[k][m]);               // exact
            }
        }
        std::vector<uint32_t> tab(3 * ntp4, 0);
        for (int k = 0; k < 3; ++k)
            for (size_t m = 0; m < ntp; ++m) tab[k * ntp4 + m] = part[k][m] | (part[k][m + 1] << 16);
        FIRCHK(hipMalloc((void **)&s.taps_split, tab.size() * sizeof(uint32_t)));
        FIRCHK(hipMemcpy(s.taps_split, tab.data(), tab.size() * sizeof(uint32_t), hipMemcpyHostToDevice));
        FIRCHK(hipFuncSetAttribute((const void *)fir_split_kernel<false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)split_lds_bytes(n_taps)));
        FIRCHK(hipFuncSetAttribute((const void *)fir_split_kernel<true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)split_lds_bytes(n_taps)));
    }
    // two-part f16 tables: each padded tap times 2^p as f16 hi + f16 lo (22 significant bits), stored as pairs (m, m + 1); p puts
    // the largest tap into [2^14, 2^15).  Needs the bf16 tables too (its second pass) and an unscale factor that is a normal f32.
    {
        float tmax = 0.0f;
        for (float t : t32) tmax = std::max(tmax, std::fabs(t));
        const int E = tmax > 0.0f ? std::ilogb(tmax) : 0;
        if (s.taps_split && tmax > 0.0f && E >= -96 && E <= 126 && half_lds_bytes(n_taps) <= LDS_PER_CU) {
            const float P = std::ldexp(1.0f, 14 - E);
            s.half_unscale = std::ldexp(1.0f, E - 28);
            const size_t ntp = t32.size(), ntp4 = (ntp + 3) & ~(size_t)3;
            std::vector<uint32_t> part[2];
            for (auto &v : part) v.assign(ntp + 1, 0);
            for (size_t m = 0; m < ntp; ++m) {
                const float hs = t32[m] * P;               // exact (a power of two; tiny taps may go subnormal: below every bar)
                const _Float16 hh = (_Float16)hs;
                const _Float16 hl = (_Float16)(hs - (float)hh);
                uint16_t b0, b1;
                memcpy(&b0, &hh, 2);
                memcpy(&b1, &hl, 2);
                part[0][m] = b0;
                part[1][m] = b1;
            }
            std::vector<uint32_t> tab(2 * ntp4, 0);
            for (int k = 0; k < 2; ++k)
                for (size_t m = 0; m < ntp; ++m) tab[k * ntp4 + m] = part[k][m] | (part[k][m + 1] << 16);
            FIRCHK(hipMalloc((void **)&s.taps_half, tab.size() * sizeof(uint32_t)));
            FIRCHK(hipMemcpy(s.taps_half, tab.data(), tab.size() * sizeof(uint32_t), hipMemcpyHostToDevice));
            FIRCHK(hipFuncSetAttribute((const void *)fir_half_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)half_lds_bytes(n_taps)));
            FIRCHK(hipFuncSetAttribute((const void *)fir_halfp_kernel<12>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)half_lds_bytes(n_taps)));
            FIRCHK(hipFuncSetAttribute((const void *)fir_halfp_kernel<16>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)half_lds_bytes(n_taps)));
        }
    }
    // DSPFX_FIR_KERNEL: 0 = exact f64 VALU kernel, 1 = MFMA; default MFMA unless the filter is tiny
    s.kernel = s.env.kernel >= 0 ? s.env.kernel : (n_taps >= 16 ? 1 : 0);
    const size_t lds = tap_table_bytes(n_taps);
    if (lds > LDS_PER_CU - 1024) s.kernel = 0;         // tap table must fit the CU's LDS
    if (s.kernel == 1) {
        const size_t sk4 = skew_lds_bytes(n_taps, 4), sk2 = skew_lds_bytes(n_taps, 2);
        if (lds > 64 * 1024) {
            FIRCHK(hipFuncSetAttribute((const void *)fir_mfma_kernel<false, 4>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
            FIRCHK(hipFuncSetAttribute((const void *)fir_mfma_kernel<true, 4>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
            FIRCHK(hipFuncSetAttribute((const void *)fir_mfma_kernel<false, 2>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
            FIRCHK(hipFuncSetAttribute((const void *)fir_mfma_kernel<true, 2>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        }
        if (sk4 > 64 * 1024 && sk4 <= LDS_PER_CU) {
            FIRCHK(hipFuncSetAttribute((const void *)fir_skew_kernel<4, 1>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)sk4));
            FIRCHK(hipFuncSetAttribute((const void *)fir_skew_kernel<4, 5>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)sk4));
        }
        if (sk2 > 64 * 1024 && sk2 <= LDS_PER_CU)
            FIRCHK(hipFuncSetAttribute((const void *)fir_skew_kernel<2, 1>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)sk2));
    }
    // the two-part f16 sweep can serve this filter: it gets its own packed copy of the history (FirState::ringh).  A ring that
    // already holds samples (a tap reload) is re-packed by the next block.
    if (s.kernel == 1 && s.taps_half && s.redo) {
        if (!s.ringh) {
            const int rc = alloc_packed(s);
            if (rc) return rc;
            s.packed_ok = s.n_seen == 0;
        }
    } else {
        free_packed(s);
    }
    return 0;
}

int fir_configure(FirState &s, const double *taps_reversed, uint32_t n_taps, int mode, uint32_t N,
                  uint32_t max_frames) {
    fir_free(s);
    {   // setup time: the one place the FIR switches are read (FirEnv)
        auto env_int = [](const char *name) {
            const char *v = getenv(name);
            return v ? atoi(v) : -1;
        };
        s.env.kernel = env_int("DSPFX_FIR_KERNEL");
        s.env.scan = env_int("DSPFX_FIR_SCAN");
        s.env.njt = env_int("DSPFX_FIR_NJT");
        s.env.skew = env_int("DSPFX_FIR_SKEW");
        s.env.split = env_int("DSPFX_FIR_SPLIT");
        s.env.half = env_int("DSPFX_FIR_HALF");
        s.env.dist = env_int("DSPFX_FIR_DIST");
        s.env.packed = env_int("DSPFX_FIR_PACKED");
        s.env.slots = env_int("DSPFX_FIR_SLOTS");
    }
    s.N = N;
    s.max_frames = max_frames;
    s.mode = mode;
    s.tiles = (N + TILE_C - 1) / TILE_C;
    s.n_seen = 0;
    s.front = 0;
    s.seen_bias = 0;
    s.dq_cap = s.dq_head = 0;
    s.R = ring_rows_for(0, n_taps, max_frames);
    const size_t ring_bytes = ring_bytes_for(s.tiles, s.R);
    FIRCHK(hipMalloc((void **)&s.ring, ring_bytes));
    FIRCHK(hipMemset(s.ring, 0, ring_bytes));
    FIRCHK(hipMalloc((void **)&s.nf_time, (size_t)s.tiles * sizeof(unsigned long long)));
    FIRCHK(hipMemset(s.nf_time, 0, (size_t)s.tiles * sizeof(unsigned long long)));
    FIRCHK(hipMalloc((void **)&s.warm_acc, (size_t)N * sizeof(double)));
    FIRCHK(hipMemset(s.warm_acc, 0, (size_t)N * sizeof(double)));
    FIRCHK(hipMalloc((void **)&s.redo, (2 + 2 * (size_t)s.tiles) * sizeof(unsigned)));
    FIRCHK(hipMemset(s.redo, 0, (2 + 2 * (size_t)s.tiles) * sizeof(unsigned)));
    s.redo_parity = 0;
    s.warm_ok = true;
    return upload_taps(s, taps_reversed, n_taps);
}

// A ring of at least `need` rows that still holds the deque (sample times [front, n_seen)).
static int grow_ring(FirState &s, uint32_t need) {
    if (need <= s.R) return 0;
    const uint64_t held = s.n_seen - s.front;
    float *nr = nullptr;
    const size_t bytes = ring_bytes_for(s.tiles, need);
    FIRCHK(hipMalloc((void **)&nr, bytes));
    FIRCHK(hipMemset(nr, 0, bytes));
    if (held) {
        hipLaunchKernelGGL(fir_rebase_kernel, dim3(1024), dim3(256), 0, nullptr, s.ring, nr, s.tiles, s.R, need, (unsigned long long)s.front,
                           (unsigned long long)s.n_seen);
        FIRCHK(hipGetLastError());
    }
    FIRCHK(hipDeviceSynchronize());
    (void)hipFree(s.ring);
    s.ring = nr;
    s.R = need;
    if (s.ringh) {                                       // same geometry as the f32 ring: rebuilt from it by the next block
        const int rc = alloc_packed(s);
        if (rc) return rc;
    }
    return 0;
}

int fir_set_taps(FirState &s, const double *taps_reversed, uint32_t n_taps, int mode) {
    if (!s.ring) {
        g_fir_err = "FIR node has no state yet";
        return DSPFX_ERR_STATE;
    }
    FIRCHK(hipDeviceSynchronize());                      // blocks in flight still read the old taps / ring
    s.mode = mode;
    s.warm_ok = s.warm_ok && s.n_seen == 0;              // running sums of the old taps are worthless
    const uint64_t held = s.n_seen - s.front;            // the deque survives the reload (fir.rs:153-171 touches `taps` only)
    const int rc = grow_ring(s, ring_rows_for(held, n_taps, s.max_frames));
    if (rc) return rc;
    return upload_taps(s, taps_reversed, n_taps);
}

void fir_free(FirState &s) {
    free_packed(s);
    if (s.ring) (void)hipFree(s.ring);
    if (s.taps64) (void)hipFree(s.taps64);
    if (s.taps32) (void)hipFree(s.taps32);
    if (s.taps_split) (void)hipFree(s.taps_split);
    if (s.taps_half) (void)hipFree(s.taps_half);
    if (s.redo) (void)hipFree(s.redo);
    if (s.nf_time) (void)hipFree(s.nf_time);
    if (s.warm_acc) (void)hipFree(s.warm_acc);
    s.taps_half = nullptr;
    s.redo = nullptr;
    s.warm_acc = nullptr;
    s.warm_ok = false;
    s.ring = nullptr;
    s.taps64 = nullptr;
    s.taps32 = nullptr;
    s.taps_split = nullptr;
    s.nf_time = nullptr;
}

void fir_reset(FirState &s, hipStream_t stream) {
    // queued on the caller's stream: behind the blocks in flight there, ahead of the next one
    if (s.ring) (void)hipMemsetAsync(s.ring, 0, ring_bytes_for(s.tiles, s.R), stream);
    if (s.ringh) {
        (void)hipMemsetAsync(s.ringh, 0, ring_bytes_for(s.tiles, s.R), stream);
        (void)hipMemsetAsync(s.peaks, 0, (size_t)s.tiles * s.peak_slots * TILE_C * sizeof(float), stream);
        s.packed_ok = true;                              // an empty history, in both forms
    }
    if (s.nf_time) (void)hipMemsetAsync(s.nf_time, 0, (size_t)s.tiles * sizeof(unsigned long long), stream);
    if (s.warm_acc) (void)hipMemsetAsync(s.warm_acc, 0, (size_t)s.N * sizeof(double), stream);
    s.warm_ok = s.warm_acc != nullptr;
    s.n_seen = 0;
    s.front = 0;
    s.seen_bias = 0;
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

static void fir_mixpipe_standalone(const FirMixPipe &mp, uint32_t nframes, hipStream_t stream) {
    if (mp.stage & 2) {
        launch_mix_reduce_final(mp.prev_b, mp.mix, nframes, stream);
        if (mp.div != 0.0f) launch_mix_finish(mp.mix, nframes, mp.div, stream);
    }
    if (mp.stage & 1) launch_mix_reduce_slices(mp.prev_a, mp.cur_b, nframes, mp.rows_a, stream);
}

int fir_process(FirState &s, const float *in, float *out, uint32_t nframes, int hop, float hop_div,
                const Layout &lay, hipStream_t stream, hipEvent_t ev_begin, hipEvent_t ev_end, float *mixpart, const FirMixPipe *mixpipe) {
    if (nframes > s.max_frames) {
        g_fir_err = "nframes > max_frames";
        return DSPFX_ERR_INVALID;
    }
    // fir.rs:187-190
    const float divisor = s.mode == DSPFX_FIR_AVERAGE ? 1.0f / (float)s.T : 1.0f;
    // (An MFMA kernel that appends the block itself -- its newest rows read from `in` -- saved the 43 us append pass but
    // ran 0.10 ms longer per block at config 4: profiles/r02_fir.txt.  Removed; the append stays a pass of its own.)
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
        ex.tiles = s.tiles;
        ex.n0 = n0;
        ex.front0 = front0;
        ex.t_lo = (long long)(front0 + ex.dfront[0]);
        ex.divisor = divisor;
        ex.mixpart = mixpart ? mixpart + f0 : nullptr;       // rows of nframes floats; this slice starts at column f0
        ex.mix_ld = nframes;
        ex.lay = lay;
        const bool steady = len0 + 1 >= s.T;                 // the first output already sees T samples
        const uint64_t d = len0 > s.T ? len0 - s.T : 0;      // deque longer than the taps: a pure extra delay (fir.rs:195-197 pops one per step)
        const bool mfma = s.kernel == 1;
        // Which sweep will this node's steady state take?  The two-part f16 sweep keeps a packed copy of the history that only
        // ITS append pass maintains; while another sweep is selected the cheaper append runs and the copy goes stale.
        const bool want_split = s.precision == DSPFX_FIR_PRECISION_SPLIT || (s.precision == DSPFX_FIR_PRECISION_DEFAULT && s.env.split != 0);
        const bool want_half = s.precision == DSPFX_FIR_PRECISION_HALF || (s.precision == DSPFX_FIR_PRECISION_DEFAULT && want_split && s.env.half != 0);
        const bool packed = want_half && s.kernel == 1 && s.ringh && s.taps_half && s.taps_split && s.redo;
        if (packed && !s.packed_ok) {                    // after an import / re-base / unpark / a spell on another sweep: rebuild from the f32 ring
            const uint64_t t_lo = n0 > (uint64_t)s.R ? n0 - s.R : 0;
            if (n0 > t_lo) {
                hipLaunchKernelGGL(fir_repack_kernel, dim3((s.N + 255) / 256, (unsigned)std::min<uint64_t>(64, (n0 - t_lo + EPOCH - 1) / EPOCH + 1)), dim3(256), 0,
                                   stream, s.ring, s.ringh, s.peaks, s.N, s.R, s.peak_slots, (unsigned long long)t_lo, (unsigned long long)n0);
            }
            s.packed_ok = true;
        }
        if (packed) {
            FirAppend2Args ap{in_s, s.ring, s.ringh, s.peaks, s.nf_time, s.N, nf, s.R, s.peak_slots, (unsigned long long)n0, hop, hop_div, lay};
            const unsigned n_epochs = (unsigned)((n0 + nf - 1) / EPOCH - n0 / EPOCH + 1);
            hipLaunchKernelGGL(fir_append2_kernel, dim3((s.N + APPEND2_CH - 1) / APPEND2_CH, n_epochs), dim3(256), 0, stream, ap);
        } else {
            s.packed_ok = false;
            const uint32_t row0 = (uint32_t)(n0 % s.R);
            hipLaunchKernelGGL(fir_append_kernel, dim3((s.N + 255) / 256, append_pieces(row0, nf)), dim3(256), 0, stream, in_s, s.ring,
                               s.nf_time, s.N, nf, row0, s.R, (unsigned long long)n0, hop, hop_div, lay);
        }
        if (ev_begin && !ev_open) {
            (void)hipEventRecord(ev_begin, stream);
            ev_open = true;
        }
        const dim3 ex_grid(s.tiles, (nf + 7) / 8);
        // the fill phase of a deque that started empty: running sums (exact); DSPFX_FIR_SCAN=0 keeps the warm-up sweep (A/B, tests)
        const bool fill = mfma && s.warm_ok && front0 == 0 && s.front == 0 && n0 + nf <= s.T && s.env.scan != 0;
        if (!steady && !fill) s.warm_ok = false;             // a filling slice the running sums did not see
        if (fill) {
            s.last_kernel = "fir_warm_scan_kernel";
            hipLaunchKernelGGL(fir_warm_scan_kernel, dim3((s.N + 255) / 256), dim3(256), 0, stream, in_s, out_s, s.warm_acc, s.taps64, s.N, nf,
                               (uint32_t)n0, hop, hop_div, divisor, ex.mixpart, nframes, lay);
            if (mixpipe && mixpipe->stage && f0 == 0) fir_mixpipe_standalone(*mixpipe, nframes, stream);
            if (ev_end && f0 + SLICE >= nframes) (void)hipEventRecord(ev_end, stream);
            continue;
        }
        if (mfma) {
            FirMfmaArgs a{};
            a.ring = s.ring;
            a.taps = s.taps32;
            a.taps_split = s.taps_split;
            a.out = out_s;
            a.nf_time = s.nf_time;
            a.N = s.N;
            a.nframes = nf;
            a.T = s.T;
            a.R = s.R;
            a.n0 = (long long)n0;
            const long long oldest = (long long)n0 - (long long)d - (long long)(s.T - 1);   // oldest sample of output 0; may be negative
            a.koff = (uint32_t)(((oldest % KC) + KC) % KC);  // the sweep starts on the 16-row group that holds it
            a.t_k0 = oldest - (long long)a.koff;
            a.rb = (uint32_t)(((a.t_k0 % (long long)s.R) + (long long)s.R) % (long long)s.R);
            a.kvalid = a.koff + s.T - 1 + nf;
            a.kpad = (a.kvalid + KC - 1) / KC * KC;
            a.tfront = (long long)front0;
            a.divisor = divisor;
            a.mixpart = ex.mixpart;
            a.mix_ld = nframes;
            a.lay = lay;
            const bool mp_pending = mixpipe && mixpipe->stage && f0 == 0;      // the block's first slice carries it
            // Tiles per wave: four (one wave sweeps the whole 128-frame slice: every history row is loaded once) unless the
            // slice is at most 64 frames, which is one wave's two tiles.  DSPFX_FIR_NJT=2|4 forces either.  (With the
            // rectangular sweep two tiles per wave were faster up to 384 taps -- a narrower band; the skewed sweep has no
            // band overhead and four tiles win at every length: profiles/r02_fir.txt.)
            const bool two = nf <= 64 || s.env.njt == 2;
            // DSPFX_FIR_SKEW=0: the rectangular sweep in steady state too (A/B, cross-checks)
            // (the skewed kernel wants its workgroup's LDS twice per CU for NJT = 4 -- two waves per SIMD -- else the rectangular sweep serves)
            const size_t lds_skew = skew_lds_bytes(s.T, two ? 2 : 4);
            const bool skew = steady && s.env.skew != 0 && lds_skew * 2 <= LDS_PER_CU;
            const unsigned grid = two ? (s.tiles + 1) / 2 : (s.tiles + 3) / 4;
            const size_t lds = skew ? lds_skew : tap_table_bytes(s.T);
            // split precision (bf16 x 3 on the bf16 matrix pipe): whole 128-frame slices in steady state.  The DEFAULT since round
            // 3 (as accurate as the f32 sweep against the f64 oracle -- 2.9e-7 vs 3.3e-7 relative RMS at 4096 taps -- bit-exact on
            // integer data, x 1.5); dspfx_set_fir_precision(F32) or DSPFX_FIR_SPLIT=0 in the environment (for nodes left at the
            // default) select the f32 sweep.
            // two-part f16 (three f16 products per term + a bf16 x 3 second pass over the tiles it lists): the DEFAULT since round 4
            // -- same bar, bit-exact on data that is exact in f16's 22 bits, config 4 1.33 -> see DESIGN 5; DSPFX_FIR_HALF=0 (for
            // nodes left at the default) or dspfx_set_fir_precision(SPLIT / F32) select the others.
            const bool half = steady && nf > 64 && s.taps_half && s.taps_split && s.redo && want_half;
            const bool split = !half && steady && nf > 64 && s.taps_split && want_split;
            s.last_kernel = !steady ? "fir_mfma_kernel<warm>" : half ? (packed ? "fir_halfp_kernel" : "fir_half_kernel") : split ? "fir_split_kernel" : skew ? "fir_skew_kernel" : "fir_mfma_kernel";
            const unsigned grid_sweep = (split || half) ? (s.tiles + SPLIT_WAVES - 1) / SPLIT_WAVES : grid;
            if (mp_pending) {
                if (steady && (split || skew || half) && grid_sweep > MIX_SLICES) a.mp = *mixpipe;     // rides in the sweep's first workgroups
                else fir_mixpipe_standalone(*mixpipe, nframes, stream);
            }
            if (!steady) {
                if (two) hipLaunchKernelGGL((fir_mfma_kernel<true, 2>), dim3(grid), dim3(256), lds, stream, a);
                else hipLaunchKernelGGL((fir_mfma_kernel<true, 4>), dim3(grid), dim3(256), lds, stream, a);
            } else if (half) {
                const int par = s.redo_parity;
                s.redo_parity ^= 1;
                a.taps_half = s.taps_half;
                a.half_unscale = s.half_unscale;
                a.redo_count = s.redo + par;
                a.redo_tiles = s.redo + 2 + (size_t)par * s.tiles;
                a.redo_clear = s.redo + (par ^ 1);
                if (packed) {
                    a.ringh = s.ringh;
                    a.peaks = s.peaks;
                    a.peak_slots = s.peak_slots;
                    if (s.env.slots == 16) hipLaunchKernelGGL(fir_halfp_kernel<16>, dim3(grid_sweep), dim3(64 * SPLIT_WAVES), half_lds_bytes(s.T), stream, a);
                    else hipLaunchKernelGGL(fir_halfp_kernel<12>, dim3(grid_sweep), dim3(64 * SPLIT_WAVES), half_lds_bytes(s.T), stream, a);
                } else {
                    hipLaunchKernelGGL(fir_half_kernel, dim3(grid_sweep), dim3(64 * SPLIT_WAVES), half_lds_bytes(s.T), stream, a);
                }
                if (ev_end && f0 + SLICE >= nframes) {      // (the dominant kernel ends here; the second pass is usually empty)
                    (void)hipEventRecord(ev_end, stream);
                    ev_end = nullptr;
                }
                a.mp = FirMixPipe{};                        // hosted by the first pass
                a.only_count = a.redo_count;
                a.only_tiles = a.redo_tiles;
                hipLaunchKernelGGL(fir_split_kernel<true>, dim3(grid_sweep), dim3(64 * SPLIT_WAVES), split_lds_bytes(s.T), stream, a);
            } else if (split) {
                hipLaunchKernelGGL(fir_split_kernel<false>, dim3((s.tiles + SPLIT_WAVES - 1) / SPLIT_WAVES), dim3(64 * SPLIT_WAVES), split_lds_bytes(s.T), stream, a);
            } else if (skew) {
                // history chunks requested 5 iterations before their first use (11 in flight or in use per wave); with 1
                // the sweep waits for HBM: 1.997 vs 1.949 ms at config 4, 3: 1.986, 7 / 9: 1.962-1.979 (DSPFX_FIR_DIST=1 for A/B)
                if (two) hipLaunchKernelGGL((fir_skew_kernel<2, 1>), dim3(grid), dim3(256), lds, stream, a);
                else if (s.env.dist == 1) hipLaunchKernelGGL((fir_skew_kernel<4, 1>), dim3(grid), dim3(256), lds, stream, a);
                else hipLaunchKernelGGL((fir_skew_kernel<4, 5>), dim3(grid), dim3(256), lds, stream, a);
            } else {
                if (two) hipLaunchKernelGGL((fir_mfma_kernel<false, 2>), dim3(grid), dim3(256), lds, stream, a);
                else hipLaunchKernelGGL((fir_mfma_kernel<false, 4>), dim3(grid), dim3(256), lds, stream, a);
            }
            if (ev_end && f0 + SLICE >= nframes) (void)hipEventRecord(ev_end, stream);
            // tiles holding a non-finite sample inside this slice's window are redone exactly (nothing to do otherwise)
            ex.only_dirty = 1;
            hipLaunchKernelGGL(fir_exact_kernel, dim3(std::min<uint32_t>(s.tiles, 512), 1), dim3(256), 0, stream, ex);
        } else {
            if (mixpipe && mixpipe->stage && f0 == 0) fir_mixpipe_standalone(*mixpipe, nframes, stream);
            ex.only_dirty = 0;
            hipLaunchKernelGGL(fir_exact_kernel, ex_grid, dim3(256), 0, stream, ex);
            if (ev_end && f0 + SLICE >= nframes) (void)hipEventRecord(ev_end, stream);
        }
    }
    FIRCHK(hipGetLastError());
    return 0;
}

// Exported state = the reference's `state: VecDeque<f64>` (fir.rs:64-65) as it stands: a 32-byte header
//   u64 n_seen   samples pushed since the deque was last empty (informational)
//   u64 held     the deque's length: < T while it fills, T in steady state, > T after a reload with a shorter
//                impulse response (fir.rs:153-171 keeps `state`; fir.rs:193-197 pops one sample per step)
//   u32 dq_cap, u32 dq_head   std VecDeque's buffer capacity and head index: they decide where as_slices() splits the
//                deque, i.e. which terms go into the partial sums `a` and `b` (fir.rs:201-216)
//   u32 n_taps, u32 reserved
// then the `held` samples [held][N] f32, oldest first (f32 is exact: the reference widens f32 samples).
size_t fir_state_bytes(const FirState &s) { return FIR_STATE_HEADER + (size_t)(s.n_seen - s.front) * s.N * sizeof(float); }

// dense rows [t0, t0 + n) <-> host, through a device staging buffer in pieces of at most 64 MiB
static int copy_rows(FirState &s, long long t0, uint64_t n, char *host, bool to_host) {
    const size_t row = (size_t)s.N * sizeof(float);
    const uint64_t per = std::max<uint64_t>(1, std::min<uint64_t>(n, ((size_t)64 << 20) / row));
    float *stage = nullptr;
    FIRCHK(hipMalloc((void **)&stage, (size_t)per * row));
    int rc = 0;
    for (uint64_t k = 0; k < n && rc == 0; k += per) {
        const uint32_t m = (uint32_t)std::min<uint64_t>(per, n - k);
        const dim3 grid((s.N + 255) / 256, std::min<uint32_t>(m, 256));
        hipError_t e = hipSuccess;
        if (!to_host) e = hipMemcpy(stage, host + (size_t)k * row, (size_t)m * row, hipMemcpyHostToDevice);
        if (e == hipSuccess) {
            hipLaunchKernelGGL(fir_rows_kernel, grid, dim3(256), 0, nullptr, s.ring, stage, s.N, s.R, t0 + (long long)k, m, to_host ? 1 : 0);
            e = hipGetLastError();
        }
        if (e == hipSuccess) e = hipDeviceSynchronize();
        if (e == hipSuccess && to_host) e = hipMemcpy(host + (size_t)k * row, stage, (size_t)m * row, hipMemcpyDeviceToHost);
        if (e != hipSuccess) {
            g_fir_err = std::string("FIR state copy: ") + hipGetErrorString(e);
            rc = DSPFX_ERR_HIP;
        }
    }
    (void)hipFree(stage);
    return rc;
}

int fir_state_export(FirState &s, void *host_dst) {
    const uint64_t held = s.n_seen - s.front;
    const uint32_t hdr32[4] = {s.dq_cap, s.dq_head, s.T, 0};
    const uint64_t seen = s.n_seen + s.seen_bias;
    memcpy(host_dst, &seen, 8);
    memcpy((char *)host_dst + 8, &held, 8);
    memcpy((char *)host_dst + 16, hdr32, 16);
    if (!held) return 0;
    FIRCHK(hipDeviceSynchronize());
    return copy_rows(s, (long long)s.front, held, (char *)host_dst + FIR_STATE_HEADER, true);
}

int64_t fir_state_import_bytes(const FirState &s, const void *host_src, size_t size) {
    if (size < FIR_STATE_HEADER) return -1;
    uint64_t held;
    memcpy(&held, (const char *)host_src + 8, 8);
    if (held > ((uint64_t)1 << 32)) return -1;
    return (int64_t)(FIR_STATE_HEADER + held * s.N * sizeof(float));
}

int fir_state_import(FirState &s, const void *host_src) {
    uint64_t seen, held;
    uint32_t hdr32[4];
    memcpy(&seen, host_src, 8);
    memcpy(&held, (const char *)host_src + 8, 8);
    memcpy(hdr32, (const char *)host_src + 16, 16);
    const char *src = (const char *)host_src + FIR_STATE_HEADER;
    FIRCHK(hipDeviceSynchronize());
    // time is re-based so that the imported deque occupies the sample times [0, held)
    s.n_seen = 0;
    s.front = 0;
    {
        const int rc = grow_ring(s, ring_rows_for(held, s.T, s.max_frames));
        if (rc) return rc;
    }
    FIRCHK(hipMemset(s.ring, 0, ring_bytes_for(s.tiles, s.R)));
    FIRCHK(hipMemset(s.nf_time, 0, (size_t)s.tiles * sizeof(unsigned long long)));
    if (held) {
        const int rc = copy_rows(s, 0, held, const_cast<char *>(src), false);
        if (rc) return rc;
        // non-finite samples of the imported history must raise their tiles' flags like appended ones do
        hipLaunchKernelGGL(fir_flag_scan_kernel, dim3(s.tiles), dim3(256), 0, nullptr, s.ring, s.nf_time, s.N, s.R, (unsigned long long)held);
        FIRCHK(hipGetLastError());
    }
    s.n_seen = held;
    s.packed_ok = false;                                 // the packed copy is rebuilt from the imported rows by the next block
    // the VecDeque's own bookkeeping: as exported when it is consistent with the length, else that of a deque that was
    // pushed `held` times from empty
    uint32_t cap = hdr32[0], head = hdr32[1];
    const bool cap_ok = cap >= held && cap >= (held ? 4u : 0u) && (cap & (cap - 1)) == 0 && (cap == 0 ? head == 0 : head < cap);
    if (cap_ok) {
        s.dq_cap = cap;
        s.dq_head = head;
    } else {
        s.dq_cap = s.dq_head = 0;
        const uint32_t T0 = s.T;
        s.T = 0xffffffffu;                                 // no pops while rebuilding
        s.n_seen = 0;
        for (uint64_t k = 0; k < held; ++k) (void)deque_step(s);
        s.T = T0;
    }
    FIRCHK(hipMemset(s.warm_acc, 0, (size_t)s.N * sizeof(double)));
    s.warm_ok = held == 0;                               // imported history: its running sums are not known
    s.seen_bias = seen >= held ? seen - held : 0;
    FIRCHK(hipDeviceSynchronize());
    return 0;
}

// ---- placement tuning runs real blocks through the node: park what they overwrite, put it back afterwards ---------
int fir_park(FirState &s, uint32_t nframes, hipStream_t stream, FirPark &p) {
    p.n_seen = s.n_seen;
    p.front = s.front;
    p.dq_cap = s.dq_cap;
    p.dq_head = s.dq_head;
    p.warm_ok = s.warm_ok;
    p.last_kernel = s.last_kernel;
    p.nframes = nframes;
    FIRCHK(hipMalloc((void **)&p.rows, (size_t)nframes * s.N * sizeof(float)));
    FIRCHK(hipMalloc((void **)&p.nf, (size_t)s.tiles * sizeof(unsigned long long)));
    FIRCHK(hipMalloc((void **)&p.acc, (size_t)s.N * sizeof(double)));
    // the rows the next `nframes` samples land in still hold the oldest part of the ring (sample times n_seen - R ...)
    hipLaunchKernelGGL(fir_rows_kernel, dim3((s.N + 255) / 256, std::min<uint32_t>(nframes, 256)), dim3(256), 0, stream, s.ring, p.rows, s.N, s.R,
                       (long long)s.n_seen + (long long)s.R * 4, nframes, 1);   // + 4 R: the same rows, a time that is never negative
    FIRCHK(hipGetLastError());
    FIRCHK(hipMemcpyAsync(p.nf, s.nf_time, (size_t)s.tiles * sizeof(unsigned long long), hipMemcpyDeviceToDevice, stream));
    FIRCHK(hipMemcpyAsync(p.acc, s.warm_acc, (size_t)s.N * sizeof(double), hipMemcpyDeviceToDevice, stream));
    return 0;
}
void fir_rewind(FirState &s, const FirPark &p) {
    s.n_seen = p.n_seen;
    s.front = p.front;
    s.dq_cap = p.dq_cap;
    s.dq_head = p.dq_head;
    s.warm_ok = p.warm_ok;
    s.last_kernel = p.last_kernel;
}
int fir_unpark(FirState &s, FirPark &p, hipStream_t stream) {
    fir_rewind(s, p);
    if (!p.rows) return 0;
    s.packed_ok = false;                                 // the probes' appends wrote their samples into the packed copy too
    hipLaunchKernelGGL(fir_rows_kernel, dim3((s.N + 255) / 256, std::min<uint32_t>(p.nframes, 256)), dim3(256), 0, stream, s.ring, p.rows, s.N, s.R,
                       (long long)s.n_seen + (long long)s.R * 4, p.nframes, 0);
    FIRCHK(hipGetLastError());
    FIRCHK(hipMemcpyAsync(s.nf_time, p.nf, (size_t)s.tiles * sizeof(unsigned long long), hipMemcpyDeviceToDevice, stream));
    FIRCHK(hipMemcpyAsync(s.warm_acc, p.acc, (size_t)s.N * sizeof(double), hipMemcpyDeviceToDevice, stream));
    return 0;
}
void fir_park_free(FirPark &p) {
    if (p.rows) (void)hipFree(p.rows);
    if (p.nf) (void)hipFree(p.nf);
    if (p.acc) (void)hipFree(p.acc);
    p.rows = nullptr;
    p.nf = nullptr;
    p.acc = nullptr;
}

const char *fir_kernel_name(const FirState &s) {
    return s.kernel != 1 ? "fir_exact_kernel" : s.last_kernel ? s.last_kernel : "fir_skew_kernel";
}

}  // namespace dspfx
