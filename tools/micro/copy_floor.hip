// What a kernel that does NOTHING but move the 3-node chain's algorithmic bytes achieves at each engine size: two streams in
// (the block's samples, the delay taps), two streams out (the block's output, the ring rows) of N x 128 f32 each, 16 bytes per lane
// per access, as many workgroups as the chip holds or one access per thread -- whichever is faster is reported.  The chain kernels'
// fractions of the 8 TB/s peak are to be read against this (round 4).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <algorithm>
typedef float f4 __attribute__((ext_vector_type(4)));
__global__ void __launch_bounds__(256) copy2(const f4 *__restrict__ a, const f4 *__restrict__ b, f4 *__restrict__ c, f4 *__restrict__ d, size_t n4) {
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (size_t)gridDim.x * 256) {
        const f4 x = __builtin_nontemporal_load(a + i), y = __builtin_nontemporal_load(b + i);
        __builtin_nontemporal_store(x + y, c + i);
        __builtin_nontemporal_store(x - y, d + i);
    }
}
// a wave walks rows like the chain kernel does: 64 lanes x 4 B x 8 rows per chunk, row stride 1 KiB (the tiled-256 layout)
__global__ void __launch_bounds__(256) copy_rows(const float *__restrict__ a, const float *__restrict__ b, float *__restrict__ c, float *__restrict__ d, unsigned N) {
    const unsigned ch = blockIdx.x * 256 + threadIdx.x;
    if (ch >= N) return;
    const size_t base = (size_t)(ch >> 8) * 128 * 256 + (ch & 255);
    for (int f0 = 0; f0 < 128; f0 += 8) {
        float x[8], y[8];
#pragma unroll
        for (int f = 0; f < 8; ++f) { x[f] = a[base + (size_t)(f0 + f) * 256]; y[f] = b[base + (size_t)(f0 + f) * 256]; }
#pragma unroll
        for (int f = 0; f < 8; ++f) { c[base + (size_t)(f0 + f) * 256] = x[f] + y[f]; d[base + (size_t)(f0 + f) * 256] = x[f] - y[f]; }
    }
}
int main(int argc, char **argv) {
    const unsigned nmax = argc > 1 ? atoi(argv[1]) : 1048576;
    float *buf[4];
    for (auto &p : buf) { hipMalloc(&p, (size_t)nmax * 128 * 4); hipMemset(p, 0, (size_t)nmax * 128 * 4); }
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    printf("# channels   bytes/block(MB)   copy us (frac of 8 TB/s)   row-walk us (frac)\n");
    for (unsigned N = 16384; N <= nmax; N += (N < 262144 ? 16384 : 131072)) {
        const size_t n4 = (size_t)N * 128 / 4;
        float best[2] = {1e9f, 1e9f};
        for (int mode = 0; mode < 3; ++mode) {
            const unsigned grid = mode == 0 ? (unsigned)((n4 + 255) / 256) : mode == 1 ? 256 * 8 : (N + 255) / 256;
            for (int rep = 0; rep < 2; ++rep) {
                for (int k = 0; k < 50; ++k) {
                    if (mode < 2) hipLaunchKernelGGL(copy2, dim3(grid), dim3(256), 0, 0, (const f4 *)buf[0], (const f4 *)buf[1], (f4 *)buf[2], (f4 *)buf[3], n4);
                    else hipLaunchKernelGGL(copy_rows, dim3(grid), dim3(256), 0, 0, buf[0], buf[1], buf[2], buf[3], N);
                }
                hipEventRecord(e0, 0);
                for (int k = 0; k < 500; ++k) {
                    if (mode < 2) hipLaunchKernelGGL(copy2, dim3(grid), dim3(256), 0, 0, (const f4 *)buf[0], (const f4 *)buf[1], (f4 *)buf[2], (f4 *)buf[3], n4);
                    else hipLaunchKernelGGL(copy_rows, dim3(grid), dim3(256), 0, 0, buf[0], buf[1], buf[2], buf[3], N);
                }
                hipEventRecord(e1, 0);
                hipEventSynchronize(e1);
                float ms; hipEventElapsedTime(&ms, e0, e1);
                float &b = best[mode == 2 ? 1 : 0];
                b = std::min(b, ms * 1000.0f / 500.0f);
            }
        }
        const double bytes = (double)N * 128 * 16;
        printf("%9u   %8.1f   %7.1f (%.3f)   %7.1f (%.3f)\n", N, bytes / 1e6, best[0], bytes / best[0] / 8e6, best[1], bytes / best[1] / 8e6);
        fflush(stdout);
    }
    return 0;
}
