// A whole graph (DAG of at most GRAPH_SLOTS fusable nodes) as ONE kernel: include/dspfx.h, dspfx_graph_set.
//
// The reference evaluates a graph node by node and every link is a pipe through memory (node.rs:267-352).  Here the
// graph's wiring is compiled into the kernel at run time (hiprtc): the engine generates a `struct Prog` whose `run`
// evaluates the nodes in the order given, every node output an array of registers, every port's collect_and_average
// (node.rs:162-194) a few adds and one division on such arrays, and instantiates graph_kernel<F, CPL, Prog>.
// Per-channel filter state, delay rings, the mix bus and the launch geometry are exactly the chain kernel's
// (chain_kernels.hip.h); what changes is only where a node's inputs come from.  A block costs one read of the Input
// node's buffer and one write of the Output node's, whatever the wiring.
//
// The kernel is only ever instantiated by hiprtc (this header is found next to libdspfx.so, like
// chain_kernels.hip.h); the engine includes it for GraphArgs, and graph_kernel_check.hip keeps it compiling.
#pragma once
#include "chain_kernels.hip.h"

namespace dspfx {

// A graph may hold twice the nodes of a fused chain stage: the chain kernels' argument block stays as it is and
// the graph kernel's appends the slots beyond it.
constexpr int GRAPH_SLOTS = 16;
// A generated kernel may read up to GRAPH_IO blocks and write up to GRAPH_IO blocks (a REGION of a graph that was cut
// into several kernels: the signals crossing the cut, side inputs and control signals from other regions).  Block 0 / 1
// of the inputs are the engine's `in` / `side`, output 0 is `out`; the rest travel here.
constexpr int GRAPH_IO = 16;
struct GraphArgs {
    ChainArgs c;                              // first: the engine builds a ChainArgs and launches either kind of kernel
    SlotArgs more[GRAPH_SLOTS - MAX_SLOTS];   // slots MAX_SLOTS .. GRAPH_SLOTS-1
    const float *xin[GRAPH_IO - 2];           // input blocks 2 .. GRAPH_IO-1
    float *xout[GRAPH_IO - 1];                // output blocks 1 .. GRAPH_IO-1
};
template <int I>
__host__ __device__ __forceinline__ const SlotArgs &gslot(const GraphArgs &g) {
    if constexpr (I < MAX_SLOTS) return g.c.slot[I];
    else return g.more[I - MAX_SLOTS];
}
#define DSPFX_FOR_GSLOTS(M) M(0) M(1) M(2) M(3) M(4) M(5) M(6) M(7) M(8) M(9) M(10) M(11) M(12) M(13) M(14) M(15)

// ---- port arithmetic on register arrays ---------------------------------------------------------------
template <int F, int CPL>
__device__ __forceinline__ void g_zero(float (&r)[F][CPL]) {      // node.rs:165 / 288: the port's buffer starts zeroed
#pragma unroll
    for (int f = 0; f < F; ++f)
#pragma unroll
        for (int j = 0; j < CPL; ++j) r[f][j] = 0.0f;
}
// An unconnected port: zeros too, but zeros the compiler must not see through.  With a literal 0.0 in a node's input the
// AMDGPU backend turns `(0.0 - z) * c` and `(0.0 - z) / c` into `(-z) * c`, `(-z) / c` (a source modifier; measured with
// -fno-fast-math) -- which is -0 where the reference's arithmetic gives +0 (an unplugged high-pass into the Output node:
// tools/graph_sweep.py seed 2044).  The chain kernels never see constant inputs; here they would.
__device__ __forceinline__ float opaque_zero() {
    float z = 0.0f;
    asm volatile("" : "+s"(z));               // one scalar register; the compiler no longer knows its value
    return z;
}
template <int F, int CPL>
__device__ __forceinline__ void g_unplugged(float (&r)[F][CPL]) {
    const float z = opaque_zero();
#pragma unroll
    for (int f = 0; f < F; ++f)
#pragma unroll
        for (int j = 0; j < CPL; ++j) r[f][j] = z;
}
template <int F, int CPL>
__device__ __forceinline__ void g_fill(float (&r)[F][CPL], float x) {   // an unconnected slider port: the slider value
#pragma unroll
    for (int f = 0; f < F; ++f)
#pragma unroll
        for (int j = 0; j < CPL; ++j) r[f][j] = x;
}
template <int F, int CPL>
__device__ __forceinline__ void g_copy(float (&r)[F][CPL], const float (&s)[F][CPL]) {   // a chain hop that is switched off
#pragma unroll
    for (int f = 0; f < F; ++f)
#pragma unroll
        for (int j = 0; j < CPL; ++j) r[f][j] = s[f][j];
}
template <int F, int CPL>
__device__ __forceinline__ void g_acc(float (&r)[F][CPL], const float (&s)[F][CPL]) {   // node.rs:172-176: buf += pipe
#pragma unroll
    for (int f = 0; f < F; ++f)
#pragma unroll
        for (int j = 0; j < CPL; ++j) r[f][j] = r[f][j] + s[f][j];
}
template <int F, int CPL>
__device__ __forceinline__ void g_acc_zero(float (&r)[F][CPL]) {   // a connected pipe that carries zeros (opaque: see g_unplugged)
    const float z = opaque_zero();
#pragma unroll
    for (int f = 0; f < F; ++f)
#pragma unroll
        for (int j = 0; j < CPL; ++j) r[f][j] = r[f][j] + z;
}
template <bool FAST, int F, int CPL>
__device__ __forceinline__ void g_div(float (&r)[F][CPL], float div, double rc) {   // node.rs:189-191: buf /= 0.0001 + k
#pragma unroll
    for (int f = 0; f < F; ++f)
#pragma unroll
        for (int j = 0; j < CPL; ++j) r[f][j] = div_c<FAST>(r[f][j], div, rc);
}
template <int F, int CPL>
__device__ __forceinline__ void g_slider(float (&p)[F][CPL], float lo, float hi) {   // dsp-stuff-derive/src/lib.rs:139-146
#pragma unroll
    for (int f = 0; f < F; ++f)
#pragma unroll
        for (int j = 0; j < CPL; ++j) p[f][j] = slider_map(p[f][j], lo, hi);
}
template <int F, int CPL>
__device__ __forceinline__ void g_add(float (&v)[F][CPL], const float (&b)[F][CPL]) {   // add.rs:29-33
#pragma unroll
    for (int f = 0; f < F; ++f)
#pragma unroll
        for (int j = 0; j < CPL; ++j) v[f][j] = v[f][j] + b[f][j];
}
template <int F, int CPL>
__device__ __forceinline__ void g_mix(float (&v)[F][CPL], const float (&b)[F][CPL], float ratio) {   // mix.rs:41-46
#pragma unroll
    for (int f = 0; f < F; ++f)
#pragma unroll
        for (int j = 0; j < CPL; ++j) v[f][j] = (b[f][j] * ratio) + (v[f][j] * (1.0f - ratio));
}
template <int F, int CPL>
__device__ __forceinline__ void g_mix_mod(float (&v)[F][CPL], const float (&b)[F][CPL], const float (&ra)[F][CPL]) {
#pragma unroll
    for (int f = 0; f < F; ++f)
#pragma unroll
        for (int j = 0; j < CPL; ++j) v[f][j] = (b[f][j] * ra[f][j]) + (v[f][j] * (1.0f - ra[f][j]));
}

// ---- the kernel ---------------------------------------------------------------------------------------
// PROG (generated):  static constexpr int sigs[GRAPH_SLOTS]  node signatures (state rows to load / store)
//                    static constexpr unsigned in_mask       bit k: input block k is read (0 = `in`, 1 = `side`, 2.. = xin)
//                    static constexpr int n_out              output blocks written (>= 1; 0 = `out`, 1.. = xout)
//                    template <int F, int CPL> static void run(g, xs, ys, st, cx)   xs[k] = input block k, ys[m] = output block m
template <int F, int CPL, class PROG>
__device__ __forceinline__ void graph_chunk(const GraphArgs &g, float (&st)[GRAPH_SLOTS][4][CPL], size_t c, const WaveAddr &w,
                                            unsigned f0, int lane, MixStage &ms, int wave) {
    const ChainArgs &a = g.c;
    float xs[GRAPH_IO][F][CPL], ys[GRAPH_IO][F][CPL];
#pragma unroll
    for (int k = 0; k < GRAPH_IO; ++k) {
        if ((PROG::in_mask >> k) & 1u) {
            const float *src = k == 0 ? a.in : k == 1 ? a.side : g.xin[k >= 2 ? k - 2 : 0];
#pragma unroll
            for (int f = 0; f < F; ++f)      // rows through a buffer descriptor (chain_kernels.hip.h, load_row)
                load_row<CPL, S_IN>(row_rsrc(src + w.io_base0 + (size_t)f0 * a.ld), w.io_off, (unsigned)f * (a.ld * 4u), xs[k][f]);
        } else {
            g_zero<F, CPL>(xs[k]);
        }
    }
    const Ctx cx{c, a.N, w.io_base0, w.io_off, w.ring_base0, w.ring_off, a.ld, f0, a.hop_div, a.hop_rc, a.third_rc, nullptr, 0, true};
    PROG::template run<F, CPL>(g, xs, ys, st, cx);
#pragma unroll
    for (int m = 0; m < GRAPH_IO; ++m) {
        if (m < PROG::n_out) {
            float *dst = m == 0 ? a.out : g.xout[m >= 1 ? m - 1 : 0];
#pragma unroll
            for (int f = 0; f < F; ++f)
                store_row<CPL, S_OUT>(row_rsrc(dst + w.io_base0 + (size_t)f0 * a.ld), w.io_off, (unsigned)f * (a.ld * 4u), ys[m][f]);
        }
    }
    if (a.mixpart) mixbus_partial<F, CPL>(ms, ys[0], true, f0, lane, wave);
}

// Covers channels a.c_base + [0, a.n_launch), n_launch % (64*CPL) == 0 (whole waves), like chain_kernel.
template <int F, int CPL, class PROG>
__global__ void __launch_bounds__(WG) graph_kernel(const GraphArgs g) {
    const ChainArgs &a = g.c;
    __shared__ MixStage ms;
    if (a.mp_stage) mixpipe_prologue(a);
    const unsigned wb = work_block(a.xcd_remap);
    const unsigned tid = wb * WG + threadIdx.x;
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const size_t rel = (size_t)tid * CPL;
    if (rel >= a.n_launch) return;                 // whole-wave uniform by construction
    const size_t c = a.c_base + rel;
    float st[GRAPH_SLOTS][4][CPL];
#define DSPFX_LD(I) load_state<PROG::sigs[I], CPL, false>(gslot<I>(g), st[I], c, a.N, true);
    DSPFX_FOR_GSLOTS(DSPFX_LD)
#undef DSPFX_LD
    const WaveAddr w = wave_addr(a, c);
    unsigned f0 = 0;
    for (; f0 + F <= a.nframes; f0 += F) {
        graph_chunk<F, CPL, PROG>(g, st, c, w, f0, lane, ms, wave);
        if (a.mixpart) mixbus_after_chunk<F, CPL>(a, ms, f0);
    }
    if constexpr (F > 1)
        for (; f0 < a.nframes; ++f0) {
            graph_chunk<1, CPL, PROG>(g, st, c, w, f0, lane, ms, wave);
            if (a.mixpart) mixbus_after_chunk<1, CPL>(a, ms, f0);
        }
#define DSPFX_ST(I)                                                                              \
    if constexpr (sig_is<K_SIGNAL_GEN>(PROG::sigs[I])) signal_gen_close_block<CPL>(gslot<I>(g), st[I], a.nframes); \
    store_state<PROG::sigs[I], CPL, false>(gslot<I>(g), st[I], c, a.N, true);
    DSPFX_FOR_GSLOTS(DSPFX_ST)
#undef DSPFX_ST
    if (a.mt_tickets) mix_tail_rows(wb, wave, lane);
}

}  // namespace dspfx
