// Dynamic (interpreting) chain kernels: any chain of <= MAX_SLOTS fusable nodes.
#include "variants.h"
namespace dspfx {
static const Variant k_dyn[] = {
    DSPFX_DYN_VARIANT("dyn_f8", 8, false),
    DSPFX_DYN_VARIANT("dyn_f4", 4, false),
    DSPFX_DYN_VARIANT("dyn_f16", 16, false),
    DSPFX_DYN_VARIANT("dyn_f8_tail", 8, true),
};
const Variant *variants_dyn(int *n) { *n = (int)(sizeof(k_dyn) / sizeof(k_dyn[0])); return k_dyn; }
}  // namespace dspfx
