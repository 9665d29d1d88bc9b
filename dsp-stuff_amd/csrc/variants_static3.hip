// Static specialisations of BASELINE config 2's chain: gain -> biquad -> delay,
// with (h) and without (n) the per-hop collect_and_average scaling.
#include "variants.h"
namespace dspfx {
#define S3H sig(K_GAIN, 0, 1), sig(K_BIQUAD, 0, 1), sig(K_REVERB, 0, 1), SIG_NONE, SIG_NONE, SIG_NONE, SIG_NONE, SIG_NONE
#define S3N sig(K_GAIN, 0, 0), sig(K_BIQUAD, 0, 0), sig(K_REVERB, 0, 0), SIG_NONE, SIG_NONE, SIG_NONE, SIG_NONE, SIG_NONE
static const Variant k_s3[] = {
    DSPFX_STATIC_VARIANT("s3h_f8_c1", 3, 8, 1, S3H),
    DSPFX_STATIC_VARIANT("s3h_f8_c2", 3, 8, 2, S3H),
    DSPFX_STATIC_VARIANT("s3h_f8_c4", 3, 8, 4, S3H),
    DSPFX_STATIC_VARIANT("s3h_f16_c1", 3, 16, 1, S3H),
    DSPFX_STATIC_VARIANT("s3h_f16_c2", 3, 16, 2, S3H),
    DSPFX_STATIC_VARIANT("s3h_f4_c4", 3, 4, 4, S3H),
    DSPFX_STATIC_VARIANT("s3h_f32_c1", 3, 32, 1, S3H),   // few channels: one wave per SIMD, memory-level parallelism from F
    DSPFX_TS_VARIANT("s3h_ts32_c1", 3, 32, 1, S3H),      // few channels: four time slices per channel group (chain_ts_kernel)
    DSPFX_TS_VARIANT("s3h_ts32_c2", 3, 32, 2, S3H),
    DSPFX_TS_TAIL_VARIANT("s3h_ts32_tail", 3, 32, S3H),    // the channels a whole-wave launch leaves over, guarded
    DSPFX_STATIC_VARIANT("s3n_f8_c1", 3, 8, 1, S3N),
    DSPFX_STATIC_VARIANT("s3n_f8_c2", 3, 8, 2, S3N),
    DSPFX_STATIC_VARIANT("s3n_f8_c4", 3, 8, 4, S3N),
};
const Variant *variants_static3(int *n) { *n = (int)(sizeof(k_s3) / sizeof(k_s3[0])); return k_s3; }
}  // namespace dspfx
#ifdef DSPFX_TS_TRACE
// debug build only (tools/ts_timeline.py): the stamps of the last time-sliced launch of THIS file's kernels
extern "C" int dspfx_debug_ts_trace(unsigned long long *host, size_t count, int clear) {
    if (clear) {
        void *p = nullptr;
        if (hipGetSymbolAddress(&p, HIP_SYMBOL(dspfx::dspfx_ts_trace)) != hipSuccess) return -1;
        return hipMemset(p, 0, sizeof(dspfx::dspfx_ts_trace)) == hipSuccess ? 0 : -1;
    }
    if (hipDeviceSynchronize() != hipSuccess) return -1;
    return hipMemcpyFromSymbol(host, HIP_SYMBOL(dspfx::dspfx_ts_trace), count * sizeof(unsigned long long)) == hipSuccess ? 0 : -1;
}
extern "C" int dspfx_debug_wg_trace(unsigned long long *host, size_t count, int clear) {
    if (clear) {
        void *p = nullptr;
        if (hipGetSymbolAddress(&p, HIP_SYMBOL(dspfx::dspfx_wg_trace)) != hipSuccess) return -1;
        return hipMemset(p, 0, sizeof(dspfx::dspfx_wg_trace)) == hipSuccess ? 0 : -1;
    }
    if (hipDeviceSynchronize() != hipSuccess) return -1;
    return hipMemcpyFromSymbol(host, HIP_SYMBOL(dspfx::dspfx_wg_trace), count * sizeof(unsigned long long)) == hipSuccess ? 0 : -1;
}
#endif
