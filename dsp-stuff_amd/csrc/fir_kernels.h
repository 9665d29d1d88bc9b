// fir_kernels.h -- FIR convolution stage (nodes/fir.rs:179-225) over N channels.
//
// History lives in HBM as a ring of R = T-1+max_frames rows, f32 (the reference widens
// f32 samples to f64, so f32 storage is exact), tiled by 32 channels:
//     ring[(c / 32) * R + (t mod R)][c % 32]
// so the K = T-1+B rows one 32-channel MFMA tile needs per block form ONE contiguous
// HBM stream of 128-byte rows.  A block first appends its own samples, then every
// output is a dot product over the T most recent rows:
//   - fir_mfma_kernel : out^T[B x 32ch] = W[B x K] * H[K x 32ch] on v_mfma_f32_32x32x2_f32
//     (exact f32 FMA chain, flushed into a second accumulator every 512 terms); W is the
//     Toeplitz matrix of the taps, generated on the fly from a zero-padded table in LDS;
//   - fir_exact_kernel: sequential f64 accumulation, bit-faithful to the reference's
//     arithmetic (small T, cross-checks).  The reference's warm-up quirk (fir.rs:193-214: while the
// VecDeque holds L < T samples, state[k] pairs with taps[k]) is reproduced by the
// index map  w(m, n) = taps_rev[m - max(0, n-T+1)]  for max(0, n-T+1) <= m <= n.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "chain_kernels.hip.h"

namespace dspfx {

struct FirState {
    float *ring = nullptr;        // [ceil(N/32)][R][32]
    double *taps64 = nullptr;     // [T] reversed, as fir.rs stores them
    float *taps32 = nullptr;      // [pad_lo + T + pad_hi] zero-padded f32 copy for the MFMA path
    uint32_t T = 0, R = 0, N = 0, max_frames = 0, pad_lo = 0, pad_hi = 0, tiles = 0;
    int mode = 0;                 // 0 Balanced, 1 Average (fir.rs:187-190)
    uint64_t n_seen = 0;          // samples consumed since the history was last empty
    int kernel = 0;               // 0 = exact f64 VALU, 1 = MFMA f32
};

int fir_configure(FirState &s, const double *taps_reversed, uint32_t n_taps, int mode, uint32_t N,
                  uint32_t max_frames);
void fir_free(FirState &s);
void fir_reset(FirState &s);
// ev_begin/ev_end (optional) are recorded around the compute kernel(s) only, not the append pass
int fir_process(FirState &s, const float *in, float *out, uint32_t nframes, int hop, float hop_div,
                const Layout &lay, hipStream_t stream, hipEvent_t ev_begin = nullptr, hipEvent_t ev_end = nullptr);
size_t fir_state_bytes(const FirState &s);
int fir_state_export(FirState &s, void *host_dst);
int fir_state_import(FirState &s, const void *host_src);
const char *fir_kernel_name(const FirState &s);
const char *fir_last_error();

}  // namespace dspfx
