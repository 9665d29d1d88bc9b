// valu_rates.hip -- issue cost (shader cycles per wave64 instruction) of the vector-ALU instructions the exact constant division is
// made of, on gfx950: the f64 form (float)((double)x * rc) = v_cvt_f64_f32 + v_mul_f64 + v_cvt_f32_f64, against the f32
// fused-multiply-add form q = x*y; r = fma(-c, q, x); q' = fma(r, y, q).  One wave per workgroup, 8 independent chains so that
// the loop measures throughput, not latency.   hipcc --offload-arch=gfx950 -O3 -o valu_rates valu_rates.hip && ./valu_rates
#include <hip/hip_runtime.h>
#include <cstdio>

template <int WHICH>
__global__ void rate_kernel(float *out, long long *cycles, float seed, double rc, float c, float y) {
    float x[8];
    for (int k = 0; k < 8; ++k) x[k] = seed + k * 0.37f + threadIdx.x * 1e-3f;
    const int iters = 4096;
    const long long t0 = clock64();
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            if (WHICH == 0) x[k] = (float)((double)x[k] * rc);                                  // cvt, mul_f64, cvt
            if (WHICH == 1) { const float q = x[k] * y; const float r = __builtin_fmaf(-c, q, x[k]); x[k] = __builtin_fmaf(r, y, q); }
            if (WHICH == 2) { double d; asm volatile("v_cvt_f64_f32 %0, %1" : "=v"(d) : "v"(x[k])); x[k] = x[k] + (float)(long long)0; asm volatile("" :: "v"(d)); }
            if (WHICH == 3) { double d = (double)x[k]; asm volatile("v_mul_f64 %0, %0, %1" : "+v"(d) : "v"(rc)); asm volatile("v_mul_f64 %0, %0, %1" : "+v"(d) : "v"(rc));
                              asm volatile("v_mul_f64 %0, %0, %1" : "+v"(d) : "v"(rc)); asm volatile("v_mul_f64 %0, %0, %1" : "+v"(d) : "v"(rc)); x[k] = (float)d; }
            if (WHICH == 4) { asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(x[k]) : "v"(y), "v"(c)); }
            if (WHICH == 5) x[k] = x[k] / c;                                                       // IEEE division (the compiler's expansion)
        }
    }
    const long long t1 = clock64();
    float s = 0.0f;
    for (int k = 0; k < 8; ++k) s += x[k];
    out[blockIdx.x * 64 + threadIdx.x] = s;
    if (threadIdx.x == 0) cycles[blockIdx.x] = t1 - t0;
}

template <int W>
static void run(const char *what, int insts_per_elem) {
    float *out;
    long long *cyc;
    hipMalloc(&out, 64 * 4 * sizeof(float));
    hipMalloc(&cyc, 4 * sizeof(long long));
    for (int rep = 0; rep < 2; ++rep) hipLaunchKernelGGL(rate_kernel<W>, dim3(4), dim3(64), 0, 0, out, cyc, 0.5f, 1.0 / 1.0001, 1.0001f, (float)(1.0 / 1.0001));
    hipDeviceSynchronize();
    long long h[4];
    hipMemcpy(h, cyc, sizeof h, hipMemcpyDeviceToHost);
    const double per = (double)h[0] / (4096.0 * 8);
    printf("%-58s %7.2f cycles per element  (%d instructions: %.2f each)\n", what, per, insts_per_elem, per / insts_per_elem);
    hipFree(out);
    hipFree(cyc);
}

int main() {
    run<4>("v_fma_f32 (yardstick)", 1);
    run<0>("f64 form: cvt_f64_f32 + mul_f64 + cvt_f32_f64", 3);
    run<1>("f32 form: mul + fma + fma", 3);
    run<2>("v_cvt_f64_f32 alone (+ one v_add_f32)", 2);
    run<3>("cvt + 4 x v_mul_f64 + cvt", 6);
    run<5>("IEEE x / c as the compiler expands it", 1);
    return 0;
}
