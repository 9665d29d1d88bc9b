// Static specialisations of BASELINE config 3/5's chain:
// biquad -> distort(SoftClip) -> delay -> biquad -> gain, with (h) / without (n) hop scaling.
#include "variants.h"
namespace dspfx {
#define S5H sig(K_BIQUAD, 0, 1), sig(K_DISTORT, D_SOFT_CLIP, 1), sig(K_REVERB, 0, 1), sig(K_BIQUAD, 0, 1), sig(K_GAIN, 0, 1), SIG_NONE, SIG_NONE, SIG_NONE
#define S5N sig(K_BIQUAD, 0, 0), sig(K_DISTORT, D_SOFT_CLIP, 0), sig(K_REVERB, 0, 0), sig(K_BIQUAD, 0, 0), sig(K_GAIN, 0, 0), SIG_NONE, SIG_NONE, SIG_NONE
static const Variant k_s5[] = {
    DSPFX_STATIC_VARIANT("s5h_f8_c1", 5, 8, 1, S5H),
    DSPFX_STATIC_VARIANT("s5h_f8_c2", 5, 8, 2, S5H),
    DSPFX_STATIC_VARIANT("s5h_f8_c4", 5, 8, 4, S5H),
    DSPFX_STATIC_VARIANT("s5h_f16_c1", 5, 16, 1, S5H),
    DSPFX_STATIC_VARIANT("s5h_f16_c2", 5, 16, 2, S5H),
    DSPFX_STATIC_VARIANT("s5h_f4_c4", 5, 4, 4, S5H),
    DSPFX_STATIC_VARIANT("s5h_f32_c1", 5, 32, 1, S5H),   // few channels: one wave per SIMD, memory-level parallelism from F
    DSPFX_TS_VARIANT("s5h_ts32_c1", 5, 32, 1, S5H),      // few channels: four time slices per channel group (chain_ts_kernel)
    DSPFX_TS_VARIANT("s5h_ts32_c2", 5, 32, 2, S5H),
    DSPFX_TS_TAIL_VARIANT("s5h_ts32_tail", 5, 32, S5H),    // the channels a whole-wave launch leaves over, guarded
    DSPFX_STATIC_VARIANT("s5n_f8_c1", 5, 8, 1, S5N),
    DSPFX_STATIC_VARIANT("s5n_f8_c2", 5, 8, 2, S5N),
    DSPFX_STATIC_VARIANT("s5n_f8_c4", 5, 8, 4, S5N),
};
const Variant *variants_static5(int *n) { *n = (int)(sizeof(k_s5) / sizeof(k_s5[0])); return k_s5; }
}  // namespace dspfx
#ifdef DSPFX_TS_TRACE
// debug build only (tools/ts_timeline.py chain5): the stamps of the last time-sliced launch of THIS file's kernels
extern "C" int dspfx_debug_ts_trace5(unsigned long long *host, size_t count, int clear) {
    if (clear) {
        void *p = nullptr;
        if (hipGetSymbolAddress(&p, HIP_SYMBOL(dspfx::dspfx_ts_trace)) != hipSuccess) return -1;
        return hipMemset(p, 0, sizeof(dspfx::dspfx_ts_trace)) == hipSuccess ? 0 : -1;
    }
    if (hipDeviceSynchronize() != hipSuccess) return -1;
    return hipMemcpyFromSymbol(host, HIP_SYMBOL(dspfx::dspfx_ts_trace), count * sizeof(unsigned long long)) == hipSuccess ? 0 : -1;
}
extern "C" int dspfx_debug_wg_trace5(unsigned long long *host, size_t count, int clear) {
    if (clear) {
        void *p = nullptr;
        if (hipGetSymbolAddress(&p, HIP_SYMBOL(dspfx::dspfx_wg_trace)) != hipSuccess) return -1;
        return hipMemset(p, 0, sizeof(dspfx::dspfx_wg_trace)) == hipSuccess ? 0 : -1;
    }
    if (hipDeviceSynchronize() != hipSuccess) return -1;
    return hipMemcpyFromSymbol(host, HIP_SYMBOL(dspfx::dspfx_wg_trace), count * sizeof(unsigned long long)) == hipSuccess ? 0 : -1;
}
#endif
