// variants.h -- registry of compiled chain-kernel variants.
//
// A variant = (slot signatures, frames per chunk F, channels per lane CPL, guard).
// Static variants are fully specialised at compile time for one chain shape (all
// node switches fold away, state and coefficients sit in registers/SGPRs); the
// dynamic variant interprets any chain of up to MAX_SLOTS nodes with wave-uniform
// switches.  Both are built from the same per-node device functions.
#pragma once
#include "chain_kernels.hip.h"

namespace dspfx {

struct Variant {
    const char *name;
    int sigs[MAX_SLOTS];   // SIG_DYN in [0] => dynamic
    int n_slots;           // static: exact chain length
    int f;
    int cpl;
    bool guard;
    bool mod;              // interpreter instantiation that evaluates control ports
    bool libm;             // interpreter instantiation that includes the f64-libm nodes
    void (*launch)(const ChainArgs &, unsigned grid, unsigned block, unsigned lds_bytes, hipStream_t);
    int ts = 0;            // time-sliced kernel (chain_ts_kernel): one workgroup per 64*cpl channels, n_frames must be 4 * ts
};

template <int F, int CPL, class SL>
void launch_static(const ChainArgs &a, unsigned grid, unsigned block, unsigned lds_bytes, hipStream_t s) {
    (void)lds_bytes;
    hipLaunchKernelGGL((chain_kernel<F, CPL, SL>), dim3(grid), dim3(block), 0, s, a);
}
template <int S, int CPL, class SL, bool GUARD = false>
void launch_ts(const ChainArgs &a, unsigned grid, unsigned block, unsigned lds_bytes, hipStream_t s) {
    (void)lds_bytes;
    hipLaunchKernelGGL((chain_ts_kernel<S, CPL, SL, GUARD>), dim3(grid), dim3(block), 0, s, a);
}
template <int F, int CPL, bool GUARD, bool MOD, bool LIBM>
void launch_dyn(const ChainArgs &a, unsigned grid, unsigned block, unsigned lds_bytes, hipStream_t s) {
    hipLaunchKernelGGL((chain_dyn_kernel<F, CPL, GUARD, MOD, LIBM>), dim3(grid), dim3(block), lds_bytes, s, a);
}

// each variants_*.hip translation unit exports one of these
const Variant *variants_dyn(int *n);
const Variant *variants_static3(int *n);
const Variant *variants_static5(int *n);

#define DSPFX_STATIC_VARIANT(NAME, NSLOTS, F, CPL, ...) \
    Variant { NAME, {__VA_ARGS__}, NSLOTS, F, CPL, false, false, true, &launch_static<F, CPL, SigList<__VA_ARGS__>> }
#define DSPFX_TS_VARIANT(NAME, NSLOTS, S, CPL, ...) \
    Variant { NAME, {__VA_ARGS__}, NSLOTS, S, CPL, false, false, true, &launch_ts<S, CPL, SigList<__VA_ARGS__>>, S }
// the guarded form for the channels a whole-wave launch leaves over (guard = true AND ts != 0; one workgroup per 64 channels)
#define DSPFX_TS_TAIL_VARIANT(NAME, NSLOTS, S, ...) \
    Variant { NAME, {__VA_ARGS__}, NSLOTS, S, 1, true, false, true, &launch_ts<S, 1, SigList<__VA_ARGS__>, true>, S }
#define DSPFX_DYN_VARIANT(NAME, F, GUARD, MOD, LIBM) \
    Variant { NAME, {SIG_DYN}, 0, F, 1, GUARD, MOD, LIBM, &launch_dyn<F, 1, GUARD, MOD, LIBM> }
#define DSPFX_DYN_VARIANT_C(NAME, F, CPL, LIBM) \
    Variant { NAME, {SIG_DYN}, 0, F, CPL, false, false, LIBM, &launch_dyn<F, CPL, false, false, LIBM> }
#define DSPFX_DYN_VARIANT_MOD_C(NAME, F, CPL) \
    Variant { NAME, {SIG_DYN}, 0, F, CPL, false, true, true, &launch_dyn<F, CPL, false, true, true> }

}  // namespace dspfx
