// Dynamic (interpreting) chain kernels: any chain of <= MAX_SLOTS fusable nodes.
#include "variants.h"
namespace dspfx {
// "copy" = the empty chain, statically specialised: a known-traffic (8 B/sample) kernel with
// exactly the chain kernels' access widths, used to calibrate the PMC byte counters.
#define NONE8 SIG_NONE, SIG_NONE, SIG_NONE, SIG_NONE, SIG_NONE, SIG_NONE, SIG_NONE, SIG_NONE
static const Variant k_dyn[] = {
    DSPFX_STATIC_VARIANT("copy_f8_c1", 0, 8, 1, NONE8),
    DSPFX_STATIC_VARIANT("copy_f8_c2", 0, 8, 2, NONE8),
    DSPFX_STATIC_VARIANT("copy_f8_c4", 0, 8, 4, NONE8),
    DSPFX_DYN_VARIANT("dyn_f8", 8, false, false, false),          // arithmetic nodes only
    DSPFX_DYN_VARIANT("dyn_f4", 4, false, false, false),
    DSPFX_DYN_VARIANT("dyn_f16", 16, false, false, false),
    DSPFX_DYN_VARIANT_C("dyn_f8_c2", 8, 2, false),                // two channels per lane
    DSPFX_DYN_VARIANT_C("dyn_f4_c2", 4, 2, false),
    DSPFX_DYN_VARIANT("dyn_libm_f8", 8, false, false, true),      // + Tanh/Sin/Atan, overdrive, chebyshev
    DSPFX_DYN_VARIANT("dyn_libm_f4", 4, false, false, true),
    DSPFX_DYN_VARIANT("dyn_libm_f16", 16, false, false, true),
    DSPFX_DYN_VARIANT_C("dyn_libm_f8_c2", 8, 2, true),
    DSPFX_DYN_VARIANT("dyn_f8_tail", 8, true, false, true),
    DSPFX_DYN_VARIANT("dyn_mod_f4_tail", 4, true, true, true),
    DSPFX_DYN_VARIANT_MOD_C("dyn_mod_f8", 8, 1),                  // + control ports
    DSPFX_DYN_VARIANT_MOD_C("dyn_mod_f8_c2", 8, 2),               // (measured: f4 0.561, f8 0.470, f4_c2 0.513, f8_c2 0.389 ms)
};
const Variant *variants_dyn(int *n) { *n = (int)(sizeof(k_dyn) / sizeof(k_dyn[0])); return k_dyn; }
}  // namespace dspfx
