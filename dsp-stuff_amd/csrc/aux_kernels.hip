// aux_kernels.hip -- Fuzz, the mix-bus second stage, and the synthetic-noise fill.
// Compiled with -ffp-contract=off like every kernel of this library.
#include "aux_kernels.h"

#include <algorithm>

namespace dspfx {

__device__ __forceinline__ unsigned abs_bits(float x) { return __float_as_uint(x) & 0x7fffffffu; }

// Fuzz (distort.rs:147-171) is block-global: three maxima over the 128 frames of a channel.  A workgroup takes 64
// channels; wave q holds frames [32q, 32q+32) of each in registers (lane = channel, so every load / store is one
// 256-byte row segment) and the three maxima are combined across the four waves through LDS.  (One lane holding all
// 128 frames needed 184 VGPRs = 2 waves per SIMD and ran at 1.5 TB/s.)  In-place safe.
constexpr int FUZZ_CH = 64, FUZZ_Q = WG / FUZZ_CH, FUZZ_F = 128 / FUZZ_Q;
__global__ void __launch_bounds__(WG) fuzz_kernel(const FuzzArgs a) {
    __shared__ unsigned smax[3][FUZZ_Q][FUZZ_CH];
    const int lane = threadIdx.x & (FUZZ_CH - 1), q = __builtin_amdgcn_readfirstlane(threadIdx.x / FUZZ_CH);
    const size_t c = (size_t)blockIdx.x * FUZZ_CH + lane;
    const bool active = c < a.N;
    const size_t cc = active ? c : 0;
    // frame-independent per-lane part of the address; the frame part is wave-uniform (scalar)
    const float *pin = a.in + a.lay.at(0, cc);
    float *pout = a.out + a.lay.at(0, cc);
    const size_t ld = a.lay.ld;
    auto wg_max = [&](int which, unsigned mine) {       // max of the four waves' partial maxima, per channel
        smax[which][q][lane] = mine;
        __syncthreads();
        unsigned m = smax[which][0][lane];
#pragma unroll
        for (int k = 1; k < FUZZ_Q; ++k) m = smax[which][k][lane] >= m ? smax[which][k][lane] : m;
        return m;
    };
    for (unsigned b0 = 0; b0 < a.nframes; b0 += 128) {
        float x[FUZZ_F];
        const unsigned f0 = b0 + q * FUZZ_F;
        unsigned m = 0;
#pragma unroll
        for (int f = 0; f < FUZZ_F; ++f) {
            float t = __builtin_nontemporal_load(pin + (size_t)(f0 + f) * ld);
            if (a.hop) t = link_hop<false>(t, a.hop_div, 0.0);
            x[f] = t;
            const unsigned bts = abs_bits(t);      // max_by(total_cmp) over |x| == integer max of the bits
            m = bts >= m ? bts : m;
        }
        const float mx = __uint_as_float(wg_max(0, m));
        const double rmx = 1.0 / (double)mx;
        // level per sample (dsp-stuff-derive/src/lib.rs:135-153): a connected port maps its signal to 0..=30 and latches
        // the block's first value per channel; otherwise the latched value or the slider
        const bool has_ctl = a.ctl != nullptr;                      // wave-uniform
        const float *pctl = has_ctl ? a.ctl + a.lay.at(0, cc) : nullptr;
        float lconst = a.level;
        if (!has_ctl && a.latch_valid) lconst = a.latch[cc];
        unsigned mzb = 0;
#pragma unroll
        for (int f = 0; f < FUZZ_F; ++f) {
            float lv = lconst;
            if (has_ctl) {
                float s = __builtin_nontemporal_load(pctl + (size_t)(f0 + f) * ld);
                if (a.ctl_hop) s = link_hop<false>(s, a.hop_div, 0.0);
                lv = slider_map(s, 0.0f, 30.0f);                     // distort.rs:37 `range = 0.0..=30.0`
                if (f == 0 && q == 0 && active) a.latch[cc] = lv;     // lib.rs:148: element 0 of the 128-frame block
            }
            const float qv = div_lane(clip1(x[f] * lv), mx, rmx);        // 158
            const float e = exp_cr(-fabsf(qv));                          // q.copysign(-1.0).exp()
            const float z = -fabsf(1.0f - e);                            // (1.0 - e).copysign(-1.0)
            x[f] = z;
            const unsigned bts = abs_bits(z);
            mzb = bts >= mzb ? bts : mzb;
        }
        const float mz = __uint_as_float(wg_max(1, mzb));
        const double rmz = 1.0 / (double)mz;
        unsigned myb = 0;
#pragma unroll
        for (int f = 0; f < FUZZ_F; ++f) {
            const float y = div_lane(clip1(x[f] * mx), mz, rmz);         // 167
            x[f] = y;
            const unsigned bts = abs_bits(y);
            myb = bts >= myb ? bts : myb;
        }
        const float my = __uint_as_float(wg_max(2, myb));
        const double rmy = 1.0 / (double)my;
        if (active) {
#pragma unroll
            for (int f = 0; f < FUZZ_F; ++f) __builtin_nontemporal_store(div_lane(x[f] * mx, my, rmy), pout + (size_t)(f0 + f) * ld);   // 171
        }
        __syncthreads();                            // smax is reused by the next 128-frame block
    }
}

// Mix bus, second and third stage as stand-alone kernels (bodies in chain_kernels.hip.h, shared with the
// in-kernel pipeline).
__global__ void __launch_bounds__(WG) mix_reduce_a_kernel(const float *part, float *part2, unsigned waves, unsigned nframes) {
    mix_slice_reduce(part, part2, waves, nframes, blockIdx.x, threadIdx.x, WG);
}
__global__ void __launch_bounds__(WG) mix_reduce_b_kernel(const float *part2, float *mix, unsigned nframes) {
    mix_final_reduce(part2, mix, nframes, 0.0f, blockIdx.x * WG + threadIdx.x, gridDim.x * WG);
}

__global__ void mix_finish_kernel(float *mix, unsigned n, float div) {
    const unsigned i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) mix[i] = mix[i] / div;
}

// collect_and_average over several pipes (node.rs:162-194): the output starts zeroed, every connected pipe is
// added in link order, then one division by the f32 count 0.0001 + n.  Element-wise, 4 floats per lane.
__global__ void __launch_bounds__(WG) link_average_kernel(const LinkAvgArgs a) {
    const size_t n4 = a.count / 4;
    for (size_t i = (size_t)blockIdx.x * WG + threadIdx.x; i < n4; i += (size_t)gridDim.x * WG) {
        float acc[4] = {0.0f, 0.0f, 0.0f, 0.0f};
        for (int k = 0; k < a.n_srcs; ++k) {
            float v[4];
            load_vec<4, false, S_IN>(a.src[k] + 4 * i, v, true);
#pragma unroll
            for (int j = 0; j < 4; ++j) acc[j] = acc[j] + v[j];
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[j] = acc[j] / a.div;
        store_vec<4, false, S_OUT>(a.dst + 4 * i, acc, true);
    }
    if (blockIdx.x == 0 && threadIdx.x < a.count % 4) {   // ragged end
        const size_t i = n4 * 4 + threadIdx.x;
        float acc = 0.0f;
        for (int k = 0; k < a.n_srcs; ++k) acc = acc + a.src[k][i];
        a.dst[i] = acc / a.div;
    }
}

// SURVEY 8d generator; same integer hash as oracle/dspfx_oracle.c:orc_noise
__device__ __forceinline__ float noise1(uint32_t seed, uint32_t channel, uint32_t n_abs) {
    uint32_t h = seed ^ (channel * 0x9E3779B9u) ^ (n_abs * 0x85EBCA6Bu);
    h ^= h >> 16;
    h *= 0x85EBCA6Bu;
    h ^= h >> 13;
    h *= 0xC2B2AE35u;
    h ^= h >> 16;
    return (float)(h >> 8) * 0x1p-23f - 1.0f;
}
__global__ void __launch_bounds__(WG) noise_kernel(float *dst, unsigned N, unsigned nframes, uint32_t c0,
                                                   uint32_t n_abs0, uint32_t seed, const Layout lay) {
    const size_t total = (size_t)N * nframes;
    for (size_t i = (size_t)blockIdx.x * WG + threadIdx.x; i < total; i += (size_t)gridDim.x * WG) {
        const uint32_t f = (uint32_t)(i / N), c = (uint32_t)(i % N);
        dst[lay.at(f, c)] = noise1(seed, c0 + c, n_abs0 + f);
    }
}

// Exhaustive proof that div_c<true>(x, c, rc) == x / c for EVERY f32 bit pattern x
// (NaNs compared as a class).  2^32 inputs, a few ms.
__global__ void __launch_bounds__(WG) verify_div_kernel(float c, double rc, unsigned long long *mismatches) {
    unsigned long long bad = 0;
    const uint64_t stride = (uint64_t)gridDim.x * WG;
    for (uint64_t b = (uint64_t)blockIdx.x * WG + threadIdx.x; b < (1ull << 32); b += stride) {
        const float x = __uint_as_float((uint32_t)b);
        const float q_ieee = div_c<false>(x, c, rc);
        const float q_fast = div_c<true>(x, c, rc);
        const bool same = (__float_as_uint(q_ieee) == __float_as_uint(q_fast)) || (q_ieee != q_ieee && q_fast != q_fast);
        bad += same ? 0 : 1;
    }
    if (bad) atomicAdd(mismatches, bad);
}

// Exhaustive comparison of the fast f64 tanh / sin / atan / exp with the library path (func 0..3) and of the
// per-lane f64 division with IEEE division on 2^32 hashed (numerator, divisor) pairs (func >= 4, the value seeds the hash), all 2^32 inputs: out[0] += inputs whose
// f32 results differ, out[1] = max ulp distance seen (NaNs compared as a class).
__global__ void __launch_bounds__(WG) verify_libm_kernel(int func, unsigned long long *out) {
    unsigned long long bad = 0, worst = 0;
    const uint64_t stride = (uint64_t)gridDim.x * WG;
    for (uint64_t b = (uint64_t)blockIdx.x * WG + threadIdx.x; b < (1ull << 32); b += stride) {
        const float x = __uint_as_float((uint32_t)b);
        float a, f;
        if (func == 0) { a = tanh_lib(x); f = tanh_cr(x); }
        else if (func == 1) { a = sin_lib(x); f = sin_cr(x); }
        else if (func == 2) { a = atan_lib(x); f = atan_cr(x); }
        else if (func == 3) { a = exp_lib(x); f = exp_cr(x); }
        else {   // per-lane division: every numerator against a divisor hashed from it (2^32 pairs out of 2^64)
            uint32_t h = (uint32_t)b * 0x9E3779B9u + (uint32_t)func;
            h ^= h >> 15; h *= 0x85EBCA6Bu; h ^= h >> 13;
            const float c = __uint_as_float(h);
            a = x / c;
            f = div_lane(x, c, 1.0 / (double)c);
        }
        if (a != a && f != f) continue;
        const uint32_t ua = __float_as_uint(a), uf = __float_as_uint(f);
        if (ua == uf) continue;
        ++bad;
        // ulp distance on the monotone integer line (sign-magnitude -> offset)
        const int64_t ia = (ua >> 31) ? -(int64_t)(ua & 0x7fffffffu) : (int64_t)ua;
        const int64_t jf = (uf >> 31) ? -(int64_t)(uf & 0x7fffffffu) : (int64_t)uf;
        const unsigned long long dist = (unsigned long long)(ia > jf ? ia - jf : jf - ia);
        worst = dist > worst ? dist : worst;
        if ((a != a) != (f != f)) worst = 0xffffffffull;
    }
    if (bad) atomicAdd(&out[0], bad);
    if (worst) atomicMax(&out[1], worst);
}
int verify_libm_on_device(int func, unsigned long long *d_out, hipStream_t s) {
    hipLaunchKernelGGL(verify_libm_kernel, dim3(256 * 16), dim3(WG), 0, s, func, d_out);
    return (int)hipGetLastError();
}

int verify_divisor_on_device(float c, double rc, unsigned long long *d_count, hipStream_t s) {
    hipLaunchKernelGGL(verify_div_kernel, dim3(256 * 16), dim3(WG), 0, s, c, rc, d_count);
    return (int)hipGetLastError();
}

__global__ void __launch_bounds__(WG) ring_copy_kernel(float *const *groups, float *dense, unsigned N, unsigned W,
                                                       unsigned D, unsigned r0, unsigned nrows, int to_dense) {
    const size_t total = (size_t)N * nrows;
    for (size_t i = (size_t)blockIdx.x * WG + threadIdx.x; i < total; i += (size_t)gridDim.x * WG) {
        const unsigned k = (unsigned)(i / N), c = (unsigned)(i % N);
        unsigned r = r0 + k;
        r = r >= D ? r - D : r;
        float *p = groups[r >> 7] + ring_in_group_offset(r, c / W, W) + (c % W);
        if (to_dense) dense[i] = *p;
        else *p = dense[i];
    }
}
void launch_ring_copy(float *const *groups, float *dense, unsigned N, unsigned W, unsigned D, unsigned r0,
                      unsigned nrows, bool to_dense, hipStream_t s) {
    const size_t total = (size_t)N * nrows;
    size_t blocks = (total + WG - 1) / WG;
    if (blocks > 256 * 32) blocks = 256 * 32;
    if (blocks == 0) blocks = 1;
    hipLaunchKernelGGL(ring_copy_kernel, dim3((unsigned)blocks), dim3(WG), 0, s, groups, dense, N, W, D, r0, nrows,
                       to_dense ? 1 : 0);
}

struct PtrPack { float *p[32]; };
__global__ void table_write_kernel(float **table, unsigned first, unsigned count, PtrPack pk) {
    if (threadIdx.x < count) table[first + threadIdx.x] = pk.p[threadIdx.x];
}
void launch_table_write(float **table, unsigned first, unsigned count, float *const *ptrs, hipStream_t s) {
    for (unsigned k = 0; k < count; k += 32) {
        PtrPack pk{};
        const unsigned n = count - k < 32u ? count - k : 32u;
        for (unsigned j = 0; j < n; ++j) pk.p[j] = ptrs[k + j];
        hipLaunchKernelGGL(table_write_kernel, dim3(1), dim3(64), 0, s, table, first + k, n, pk);
    }
}

void launch_fuzz(const FuzzArgs &a, hipStream_t s) {
    hipLaunchKernelGGL(fuzz_kernel, dim3((a.N + FUZZ_CH - 1) / FUZZ_CH), dim3(WG), 0, s, a);
}
void launch_mix_reduce(const float *part, float *part2, float *mix, unsigned nframes, unsigned waves, hipStream_t s) {
    hipLaunchKernelGGL(mix_reduce_a_kernel, dim3(MIX_SLICES), dim3(WG), 0, s, part, part2, waves, nframes);
    hipLaunchKernelGGL(mix_reduce_b_kernel, dim3((nframes + WG - 1) / WG), dim3(WG), 0, s, part2, mix, nframes);
}
void launch_mix_reduce_final(const float *part2, float *mix, unsigned nframes, hipStream_t s) {
    hipLaunchKernelGGL(mix_reduce_b_kernel, dim3((nframes + WG - 1) / WG), dim3(WG), 0, s, part2, mix, nframes);
}
void launch_mix_reduce_slices(const float *part, float *part2, unsigned nframes, unsigned waves, hipStream_t s) {
    hipLaunchKernelGGL(mix_reduce_a_kernel, dim3(MIX_SLICES), dim3(WG), 0, s, part, part2, waves, nframes);
}
void launch_link_average(const LinkAvgArgs &a, hipStream_t s) {
    const size_t n4 = a.count / 4;
    const unsigned grid = (unsigned)std::min<size_t>(std::max<size_t>(1, (n4 + WG - 1) / WG), 256 * 32);
    link_average_kernel<<<grid, WG, 0, s>>>(a);
}
void launch_mix_finish(float *mix, unsigned n, float div, hipStream_t s) {
    hipLaunchKernelGGL(mix_finish_kernel, dim3((n + 127) / 128), dim3(128), 0, s, mix, n, div);
}
void launch_noise(float *dst, unsigned N, unsigned nframes, uint32_t c0, uint32_t n_abs0, uint32_t seed,
                  const Layout &lay, hipStream_t s) {
    const size_t total = (size_t)N * nframes;
    size_t blocks = (total + WG - 1) / WG;
    if (blocks > 256 * 32) blocks = 256 * 32;
    if (blocks == 0) blocks = 1;
    hipLaunchKernelGGL(noise_kernel, dim3((unsigned)blocks), dim3(WG), 0, s, dst, N, nframes, c0, n_abs0, seed, lay);
}

}  // namespace dspfx
