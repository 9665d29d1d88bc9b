// Does v_mfma_f32_32x32x16_f16 honour f16 SUBNORMAL inputs, and how does the f32 -> f16 pack round?  (round 4, FIR two-part split)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstring>
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
__global__ void k(float *out, float a, float b, const float *cv, unsigned *packed) {
    f16x8 A, B;
    for (int i = 0; i < 8; ++i) { A[i] = (_Float16)a; B[i] = (_Float16)b; }
    f32x16 c;
    for (int i = 0; i < 16; ++i) c[i] = 0.0f;
    c = __builtin_amdgcn_mfma_f32_32x32x16_f16(A, B, c, 0, 0, 0);
    if (threadIdx.x == 0) out[0] = c[0];
    if (threadIdx.x < 8) {
        const f16x2 p = __builtin_convertvector(f32x2{cv[2 * threadIdx.x], cv[2 * threadIdx.x + 1]}, f16x2);
        unsigned u;
        __builtin_memcpy(&u, &p, 4);
        packed[threadIdx.x] = u;
    }
}
int main() {
    float *d; unsigned *dp; float *dc;
    hipMalloc(&d, 4); hipMalloc(&dp, 64); hipMalloc(&dc, 64);
    // values for the pack: 1 + 2^-11 (a tie for f16: RNE -> 1.0, RTZ -> 1.0), 1 + 3*2^-12 (RNE -> 1+2^-10, RTZ -> 1.0), subnormal 2^-20, 70000 (overflow)
    float cv[16] = {1.0f + 0x1p-11f, 1.0f + 3 * 0x1p-12f, 0x1p-20f, 70000.0f, 65519.0f, 65520.0f, -0x1p-25f, 3 * 0x1p-26f, 0, 0, 0, 0, 0, 0, 0, 0};
    hipMemcpy(dc, cv, 64, hipMemcpyHostToDevice);
    const float as[3] = {0x1p-20f, 0x1p-14f, 0x1p-24f};
    for (float a : as) {
        hipLaunchKernelGGL(k, 1, 64, 0, 0, d, a, 1024.0f, dc, dp);
        float h; hipMemcpy(&h, d, 4, hipMemcpyDeviceToHost);
        printf("A = %g (f16 %s), B = 1024, K = 16: C = %g (expected %g)\n", a, a < 0x1p-14f ? "subnormal" : "normal", h, 16.0 * a * 1024.0);
    }
    unsigned hp[8]; hipMemcpy(hp, dp, 32, hipMemcpyDeviceToHost);
    for (int i = 0; i < 4; ++i) printf("pack(%a, %a) = %04x %04x\n", cv[2 * i], cv[2 * i + 1], hp[i] & 0xffff, hp[i] >> 16);
    return 0;
}
