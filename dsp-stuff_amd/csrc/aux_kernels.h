// aux_kernels.h -- launchers of the non-template kernels (aux_kernels.hip).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/dspfx.h"

#include "chain_kernels.hip.h"

namespace dspfx {

// distort.rs:146-172 (Fuzz), block-global over reference blocks of 128 frames.
struct FuzzArgs {
    const float *in;
    float *out;
    unsigned N;
    unsigned nframes;   // multiple of 128
    float level;
    float hop_div;
    int hop;
    // the `level` control port (distort.rs:176-180 maps it for every mode; fuzz zips it per sample, 154-160):
    // ctl = connected signal in the sample layout (or nullptr), latch = per-channel latched slider [N] (or nullptr)
    int ctl_hop;        // the control link's collect_and_average hop
    const float *ctl;
    float *latch;
    int latch_valid;
    int pad_;
    Layout lay;
};
void launch_fuzz(const FuzzArgs &a, hipStream_t s);
// mix[f] = fixed-order sum over waves of part[wave][f]; part2 = scratch [128][nframes]
void launch_mix_reduce(const float *part, float *part2, float *mix, unsigned nframes, unsigned waves, hipStream_t s);
// node.rs:189-191: mix[i] /= div
void launch_mix_finish(float *mix, unsigned n, float div, hipStream_t s);
void launch_mix_reduce_slices(const float *part, float *part2, unsigned nframes, unsigned waves, hipStream_t s);
void launch_mix_reduce_final(const float *part2, float *mix, unsigned nframes, hipStream_t s);

// node.rs:162-194 for a port with several connected pipes: dst = (0 + src0 + src1 + ...) / f32(0.0001 + n)
struct LinkAvgArgs {
    const float *src[DSPFX_MAX_LINKS];
    int n_srcs;
    float *dst;
    size_t count;
    float div;
};
void launch_link_average(const LinkAvgArgs &a, hipStream_t s);
// launches the exhaustive check of the fast constant division; *d_count accumulates mismatches
int verify_divisor_on_device(float c, double rc, unsigned long long *d_count, hipStream_t s);
// d_out[0] += differing inputs, d_out[1] = max ulp distance: fast tanh (func 0) / sin (func 1) vs the library path
int verify_libm_on_device(int func, unsigned long long *d_out, hipStream_t s);
// dense[k][c] <-> ring row (r0 + k) mod D of channel c, k < nrows  (state export / import)
void launch_ring_copy(float *const *groups, float *dense, unsigned N, unsigned W, unsigned D, unsigned r0,
                      unsigned nrows, bool to_dense, hipStream_t s);
// table[first + k] = ptrs[k], k < count, in stream order.  The pointers travel in the kernel arguments (32 per launch): no host
// staging buffer has to outlive the call, and no copy engine -- whose pageable-memory path waits for the stream -- is involved.
void launch_table_write(float **table, unsigned first, unsigned count, float *const *ptrs, hipStream_t s);
// dst[f][c] = noise(seed, c0 + c, n_abs0 + f)
void launch_noise(float *dst, unsigned N, unsigned nframes, uint32_t c0, uint32_t n_abs0, uint32_t seed,
                  const Layout &lay, hipStream_t s);

}  // namespace dspfx
