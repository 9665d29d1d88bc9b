// aux_kernels.hip -- Fuzz, the mix-bus second stage, and the synthetic-noise fill.
// Compiled with -ffp-contract=off like every kernel of this library.
#include "aux_kernels.h"

namespace dspfx {

__device__ __forceinline__ unsigned abs_bits(float x) { return __float_as_uint(x) & 0x7fffffffu; }

// One lane = one channel; the 128 samples of a reference block stay in registers
// across the three max-reductions (distort.rs:147-171).  In-place safe.
__global__ void __launch_bounds__(WG) fuzz_kernel(const FuzzArgs a) {
    const size_t c = (size_t)blockIdx.x * WG + threadIdx.x;
    if (c >= a.N) return;
    for (unsigned b0 = 0; b0 < a.nframes; b0 += 128) {
        float x[128];
        float lv[1];
        lv[0] = a.level;
        unsigned m = 0;
#pragma unroll
        for (int f = 0; f < 128; ++f) {
            float t = a.in[a.lay.at(b0 + f, c)];
            if (a.hop) t = link_hop<false>(t, a.hop_div, 0.0);
            x[f] = t;
            const unsigned bts = abs_bits(t);      // max_by(total_cmp) over |x| == integer max of the bits
            m = bts >= m ? bts : m;
        }
        const float mx = __uint_as_float(m);
        unsigned mzb = 0;
#pragma unroll
        for (int f = 0; f < 128; ++f) {
            const float q = clip1(x[f] * lv[0]) / mx;            // 158
            const float e = exp_cr(-fabsf(q));                     // q.copysign(-1.0).exp()
            const float z = -fabsf(1.0f - e);                    // (1.0 - e).copysign(-1.0)
            x[f] = z;
            const unsigned bts = abs_bits(z);
            mzb = bts >= mzb ? bts : mzb;
        }
        const float mz = __uint_as_float(mzb);
        unsigned myb = 0;
#pragma unroll
        for (int f = 0; f < 128; ++f) {
            const float y = clip1(x[f] * mx) / mz;               // 167
            x[f] = y;
            const unsigned bts = abs_bits(y);
            myb = bts >= myb ? bts : myb;
        }
        const float my = __uint_as_float(myb);
#pragma unroll
        for (int f = 0; f < 128; ++f) a.out[a.lay.at(b0 + f, c)] = x[f] * mx / my;   // 171
    }
}

// Deterministic fixed-order sum of the per-wave partials of one frame: every lane sums a fixed
// strided subset (4 independent accumulators keep the loads in flight), then a fixed LDS tree.
__global__ void __launch_bounds__(WG) mix_reduce_kernel(const float *part, float *mix, unsigned stride) {
    __shared__ float sh[WG];
    const unsigned f = blockIdx.x;
    const float *row = part + (size_t)f * stride;
    float a0 = 0.0f, a1 = 0.0f, a2 = 0.0f, a3 = 0.0f;
    unsigned i = threadIdx.x;
    for (; i + 3 * WG < stride; i += 4 * WG) {
        const float x0 = row[i], x1 = row[i + WG], x2 = row[i + 2 * WG], x3 = row[i + 3 * WG];
        a0 = a0 + x0; a1 = a1 + x1; a2 = a2 + x2; a3 = a3 + x3;
    }
    for (; i < stride; i += WG) a0 = a0 + row[i];
    sh[threadIdx.x] = (a0 + a1) + (a2 + a3);
    __syncthreads();
    for (int o = WG / 2; o >= 1; o >>= 1) {
        if ((int)threadIdx.x < o) sh[threadIdx.x] = sh[threadIdx.x] + sh[threadIdx.x + o];
        __syncthreads();
    }
    if (threadIdx.x == 0) mix[f] = sh[0];
}

__global__ void mix_finish_kernel(float *mix, unsigned n, float div) {
    const unsigned i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) mix[i] = mix[i] / div;
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

int verify_divisor_on_device(float c, double rc, unsigned long long *d_count, hipStream_t s) {
    hipLaunchKernelGGL(verify_div_kernel, dim3(256 * 16), dim3(WG), 0, s, c, rc, d_count);
    return (int)hipGetLastError();
}

void launch_fuzz(const FuzzArgs &a, hipStream_t s) {
    hipLaunchKernelGGL(fuzz_kernel, dim3((a.N + WG - 1) / WG), dim3(WG), 0, s, a);
}
void launch_mix_reduce(const float *part, float *mix, unsigned nframes, unsigned stride, hipStream_t s) {
    hipLaunchKernelGGL(mix_reduce_kernel, dim3(nframes), dim3(WG), 0, s, part, mix, stride);
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
