// Build-time check of graph_kernel.hip.h.  That header is otherwise compiled only at run time (hiprtc, from the
// translation unit dspfx.hip's graph_source() writes); this file spells out one such generated program by hand --
// every port form the generator emits -- so `make` fails when the header or the helpers it calls stop compiling.
// The object is not linked into libdspfx.so.
#include "graph_kernel.hip.h"

namespace dspfx {
struct CheckProg {
    static constexpr int sigs[GRAPH_SLOTS] = {sig(K_GAIN), sig(K_REVERB), sig(K_MIX), sig(K_SIGNAL_GEN, G_SINE), sig(K_OVERDRIVE),
                                              sig(K_BIQUAD), sig(K_DISTORT, D_TANH), sig(K_ADD), sig(K_LOW_PASS), SIG_NONE, SIG_NONE,
                                              SIG_NONE, SIG_NONE, SIG_NONE, SIG_NONE, SIG_NONE};
    static constexpr unsigned in_mask = 0x7;      // `in`, `side` and one extra block
    static constexpr int n_out = 2;
    template <int F, int CPL>
    static __device__ __forceinline__ void run(const GraphArgs &g, const float (&xs)[GRAPH_IO][F][CPL], float (&ys)[GRAPH_IO][F][CPL],
                                               float (&st)[GRAPH_SLOTS][4][CPL], const Ctx &cx) {
        const float (&x)[F][CPL] = xs[0];
        const float (&x2)[F][CPL] = xs[1];
        float (&y)[F][CPL] = ys[0];
        RingPre<F, CPL> pre1; ring_prefetch<F, CPL, false>(gslot<1>(g), cx, pre1);
        // node 0: one link from the Input node
        float v0[F][CPL]; g_zero<F, CPL>(v0); g_acc<F, CPL>(v0, x); g_div<true, F, CPL>(v0, 0x1.0006p+0f, 0x1.fff2p-1);
        apply_node<K_GAIN, 0, F, CPL, false, true>(gslot<0>(g), v0, st[0], cx);
        // node 1: fan-in of a node and a pipe of zeros, IEEE division
        float v1[F][CPL]; g_zero<F, CPL>(v1); g_acc<F, CPL>(v1, v0); g_acc_zero<F, CPL>(v1); g_div<false, F, CPL>(v1, 0x1.00034p+1f, 0x1.fff98p-2);
        ring_apply<F, CPL, false>(gslot<1>(g), v1, pre1, cx);
        // node 3: a generator whose frequency slider is fed by node 1
        float v3[F][CPL]; g_unplugged<F, CPL>(v3);
        float p3_0[F][CPL]; g_fill<F, CPL>(p3_0, gslot<3>(g).p[0]);
        float p3_1[F][CPL]; g_zero<F, CPL>(p3_1); g_acc<F, CPL>(p3_1, v1); g_div<true, F, CPL>(p3_1, 0x1.0006p+0f, 0x1.fff2p-1);
        g_slider<F, CPL>(p3_1, 0x1.99999ap-4f, 0x1.388p+14f);
        siggen_mod_core<G_SINE, F, CPL>(gslot<3>(g), v3, st[3], p3_0, p3_1, cx);
        // node 2: Mix with its ratio slider and "b" port connected
        float v2[F][CPL]; g_zero<F, CPL>(v2); g_acc<F, CPL>(v2, v0); g_div<true, F, CPL>(v2, 0x1.0006p+0f, 0x1.fff2p-1);
        float p2_0[F][CPL]; g_zero<F, CPL>(p2_0); g_acc<F, CPL>(p2_0, v3); g_div<true, F, CPL>(p2_0, 0x1.0006p+0f, 0x1.fff2p-1);
        g_slider<F, CPL>(p2_0, 0x0p+0f, 0x1p+0f);
        float b2[F][CPL]; g_zero<F, CPL>(b2); g_acc<F, CPL>(b2, v1); g_div<true, F, CPL>(b2, 0x1.0006p+0f, 0x1.fff2p-1);
        g_mix_mod<F, CPL>(v2, b2, p2_0);
        // node 4: one of three sliders connected
        float v4[F][CPL]; g_zero<F, CPL>(v4); g_acc<F, CPL>(v4, v2); g_div<true, F, CPL>(v4, 0x1.0006p+0f, 0x1.fff2p-1);
        float p4_0[F][CPL]; g_fill<F, CPL>(p4_0, gslot<4>(g).p[0]);
        float p4_1[F][CPL]; g_fill<F, CPL>(p4_1, gslot<4>(g).p[1]);
        float p4_2[F][CPL]; g_zero<F, CPL>(p4_2); g_acc<F, CPL>(p4_2, v3); g_div<true, F, CPL>(p4_2, 0x1.0006p+0f, 0x1.fff2p-1);
        g_slider<F, CPL>(p4_2, 0x0p+0f, 0x1p+0f);
        overdrive_mod_core<F, CPL>(v4, p4_0, p4_1, p4_2);
        float v5[F][CPL]; g_zero<F, CPL>(v5); g_acc<F, CPL>(v5, v4); g_div<true, F, CPL>(v5, 0x1.0006p+0f, 0x1.fff2p-1);
        apply_node<K_BIQUAD, 0, F, CPL, false, true>(gslot<5>(g), v5, st[5], cx);
        float v6[F][CPL]; g_zero<F, CPL>(v6); g_acc<F, CPL>(v6, v5); g_div<true, F, CPL>(v6, 0x1.0006p+0f, 0x1.fff2p-1);
        float p6_0[F][CPL]; g_zero<F, CPL>(p6_0); g_acc<F, CPL>(p6_0, v3); g_div<true, F, CPL>(p6_0, 0x1.0006p+0f, 0x1.fff2p-1);
        g_slider<F, CPL>(p6_0, 0x0p+0f, 0x1.ep+4f);
        distort_mod_core<D_TANH, F, CPL>(v6, p6_0);
        // node 7: Add with an unconnected "b" port, Gain-style slider on a plain node is gain_mod_core
        float v7[F][CPL]; g_zero<F, CPL>(v7); g_acc<F, CPL>(v7, v6); g_div<true, F, CPL>(v7, 0x1.0006p+0f, 0x1.fff2p-1);
        float b7[F][CPL]; g_unplugged<F, CPL>(b7);
        g_add<F, CPL>(v7, b7);
        gain_mod_core<F, CPL>(v7, p2_0);
        g_mix<F, CPL>(v7, b2, gslot<2>(g).p[0]);
        // node 8: a slot beyond the chain kernels' argument block
        float v8[F][CPL]; g_zero<F, CPL>(v8); g_acc<F, CPL>(v8, v7); g_div<true, F, CPL>(v8, 0x1.0006p+0f, 0x1.fff2p-1);
        apply_node<K_LOW_PASS, 0, F, CPL, false, true>(gslot<8>(g), v8, st[8], cx);
        // Output node: two links
        g_zero<F, CPL>(y); g_acc<F, CPL>(y, v8); g_acc<F, CPL>(y, x2); g_div<true, F, CPL>(y, 0x1.00034p+1f, 0x1.fff98p-2);
        // a second output block: a signal handed to the next region as it is, and an average over an extra input
        g_zero<F, CPL>(ys[1]); g_acc<F, CPL>(ys[1], v5); g_acc<F, CPL>(ys[1], xs[2]); g_div<true, F, CPL>(ys[1], 0x1.00034p+1f, 0x1.fff98p-2);
    }
};
template __global__ void graph_kernel<8, 2, CheckProg>(const GraphArgs);
template __global__ void graph_kernel<8, 1, CheckProg>(const GraphArgs);
}  // namespace dspfx
