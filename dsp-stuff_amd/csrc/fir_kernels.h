// fir_kernels.h -- FIR convolution stage (nodes/fir.rs:179-225) over N channels.
//
// History lives in HBM as a ring of R rows (R a multiple of 16), f32 (the reference widens f32 samples to f64, so f32
// storage is exact), tiled by 32 channels and, inside a tile, by groups of 16 rows stored so that the eight consecutive rows
// an MFMA lane feeds into one 16-row chunk are two 16-byte pieces (fir_kernels.hip, ring_in_tile):
//     ring[c / 32][(t mod R) / 16][row / 4 % 2][row / 8 % 2][c % 32][row % 4]
// so the K = T-1+B rows one 32-channel MFMA tile needs per block form ONE contiguous HBM stream of 2 KiB chunks.
// Every output is a dot product over the samples the reference's VecDeque holds at that step:
//   - fir_append_kernel: the block's samples (hop applied) go into the ring; non-finite ones raise the tile's flag;
//   - the sweeps: out^T[B x 32ch] = W[B x K] * H[K x 32ch] on the matrix pipe, W the Toeplitz matrix of the taps generated
//     on the fly from zero-padded tables in LDS, f32 chains flushed into separate totals every ~512 terms:
//       fir_skew_kernel   steady state, v_mfma_f32_32x32x2_f32 (exact f32 FMA chain); the four output tiles of a wave share
//                         one set of weights per iteration, each on its own history chunk out of a register window;
//       fir_mfma_kernel   the rectangular form (one history chunk per step shared by the tiles, per-tile weights): while the
//                         deque fills (WARM: the reference's front-aligned pairing) and as the steady-state fallback;
//       fir_split_kernel  f32 operands as three bf16 parts each, six v_mfma_f32_32x32x16_bf16 per 16 taps and tile;
//       fir_half_kernel   the default since round 4: f32 operands as f16 hi + f16 lo (x 2^14, taps x 2^p), THREE
//                         v_mfma_f32_32x32x16_f16 per 16 taps and tile; tiles with a channel outside f16's range (peak >= 3.998, or
//                         below 2^-13 without being silent) are listed by the sweep and redone by fir_split_kernel right behind it;
//     non-finite (and huge) samples are replaced by 0 in the MFMA operands and their tiles redone by the exact kernel;
//     when the FIR node ends the chain the sweeps' epilogue also leaves the Output node's mix-bus partials;
//   - fir_warm_scan_kernel: while a deque that started empty is still filling, state[k] pairs with taps[k] (fir.rs:193-214)
//     and nothing is popped, so output n is the PREFIX sum over samples 0..n of x[m] taps[m] -- one running f64 sum per
//     channel, the same sequence of f64 additions as the reference's fresh sum: bit-exact, and O(1) per sample where the
//     warm-up sweep costs more than a steady-state block;
//   - fir_exact_kernel: sequential f64 accumulation in deque order, split at the deque's wrap point into the
//     reference's two partial sums (`a`, `b`: fir.rs:201-216), bit-faithful; serves tiny filters, taps that do
//     not fit LDS, cross-checks, and re-computes every tile flagged non-finite after the MFMA pass.
// The deque itself is modelled on the host: `front` = absolute index of its oldest sample (so its length is
// n_seen - front: T in steady state, < T while warming up -- fir.rs:193-214 then pairs state[k] with taps[k] --
// and > T after a reload with a shorter impulse response, fir.rs:153-171 + 193-197: a pure extra delay), plus the
// ring-buffer bookkeeping of std VecDeque (capacity, head) that decides the a/b split.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "chain_kernels.hip.h"

namespace dspfx {

// The DSPFX_FIR_* switches, read ONCE when the node is configured (fir_configure: dspfx_chain_set) -- fir_process, on the per-block
// path, never calls getenv.  -1 = not set.
struct FirEnv {
    int kernel = -1;   // DSPFX_FIR_KERNEL: 0 exact f64 VALU kernel, 1 MFMA
    int scan = -1;     // DSPFX_FIR_SCAN=0: the warm-up sweep instead of the running sums while the deque fills
    int njt = -1;      // DSPFX_FIR_NJT=2|4: tiles per wave
    int skew = -1;     // DSPFX_FIR_SKEW=0: the rectangular sweep in steady state too
    int split = -1;    // DSPFX_FIR_SPLIT=0: nodes left at the default precision take the f32 sweep
    int half = -1;     // DSPFX_FIR_HALF=0: ... the bf16 x 3 sweep
    int dist = -1;     // DSPFX_FIR_DIST=1: history chunks requested one iteration ahead (A/B)
    int slots = -1;    // DSPFX_FIR_SLOTS=12|16: register slots of the packed sweep's chunk ring (A/B)
    int packed = -1;   // DSPFX_FIR_PACKED=0: the two-part f16 sweep splits the f32 history itself (round 4's form; A/B, bisecting)
};

struct FirState {
    FirEnv env;
    float *ring = nullptr;        // [ceil(N/32)] tiles of R * 32 + 32 floats
    // The two-part f16 sweep's own copy of the history (round 5): every sample ALREADY split into f16 hi + f16 lo of x 2^14, laid out
    // like `ring` (same tile stride, same 2 KiB chunks) but in MFMA operand order -- chunk = [part hi | lo][kh][channel][4 dwords],
    // a dword = the parts of two consecutive rows -- so the sweep's two 16-byte loads per lane and chunk ARE its B operands; and
    // the per-channel peak |x 2^14| of every 128 sample times (an "epoch": slot = epoch mod peak_slots), which the sweep reduces
    // over its window (<= 34 entries) instead of scanning every sample.  Written by the append pass (fir_append2_kernel); any other
    // writer of `ring` (state import, a tap reload that re-bases the ring, placement tuning's unpark) clears packed_ok and the next
    // block rebuilds both from `ring` (fir_repack_kernel).  Null when the two-part sweep cannot serve this filter.
    unsigned *ringh = nullptr;
    float *peaks = nullptr;       // [tiles][peak_slots][32]
    uint32_t peak_slots = 0;
    bool packed_ok = false;
    double *taps64 = nullptr;     // [T] reversed, as fir.rs stores them
    float *taps32 = nullptr;      // [pad_lo + T + pad_hi] zero-padded f32 copy for the MFMA path
    unsigned *taps_split = nullptr;   // split-precision sweep: [3][ntp4] bf16 pair tables of the same padded taps (or null)
    unsigned *taps_half = nullptr;    // two-part f16 sweep: [2][ntp4] f16 pair tables of the padded taps x 2^p (or null)
    float half_unscale = 0.0f;        // 2^-(14 + p): what that sweep's accumulators are multiplied by
    unsigned *redo = nullptr;         // two-part sweep: [2] counts + [2][tiles] lists of the tiles its second pass redoes, used in turn
    int redo_parity = 0;
    unsigned long long *nf_time = nullptr;   // [tiles]: 1 + absolute time of the newest non-finite sample of the tile (0: none)
    uint32_t T = 0, R = 0, N = 0, max_frames = 0, pad_lo = 0, pad_hi = 0, tiles = 0;
    int mode = 0;                 // 0 Balanced, 1 Average (fir.rs:187-190)
    uint64_t n_seen = 0;          // samples consumed since the history was last empty
    uint64_t front = 0;           // absolute index of the deque's oldest sample (deque length = n_seen - front)
    uint64_t seen_bias = 0;       // samples the deque had seen before a state import re-based time (reporting only)
    uint32_t dq_cap = 0, dq_head = 0;   // std VecDeque bookkeeping (a/b slice split of the exact kernel)
    int kernel = 0;               // 0 = exact f64 VALU, 1 = MFMA f32
    int precision = 0;            // dspfx_fir_precision: 0 default (two-part f16 / environment), 1 f32, 2 split (bf16 x 3), 3 two-part f16
    const char *last_kernel = nullptr;   // the sweep kernel of the last block (reporting)
    double *warm_acc = nullptr;   // [N] running f64 sums of the fill phase (fir_warm_scan_kernel)
    bool warm_ok = false;         // warm_acc holds the sums of everything pushed since the deque was last empty
};

// The pipelined mix bus' work of EARLIER blocks (chain_kernels.hip.h, mixpipe_prologue) for a block whose last launch is a
// FIR sweep: the sweep's first workgroups host it when it has enough of them, else fir_process launches the stand-alone
// kernels.  stage bit 0: slice-reduce prev_a into cur_b; bit 1: final-reduce prev_b into mix (/ div when non-zero).
struct FirMixPipe {
    int stage = 0;
    unsigned rows_a = 0;
    const float *prev_a = nullptr;
    float *cur_b = nullptr;
    const float *prev_b = nullptr;
    float *mix = nullptr;
    float div = 0.0f;
};

int fir_configure(FirState &s, const double *taps_reversed, uint32_t n_taps, int mode, uint32_t N,
                  uint32_t max_frames);
// Impulse-response reload (fir.rs:153-171): new taps, the history and the deque's length are KEPT.
int fir_set_taps(FirState &s, const double *taps_reversed, uint32_t n_taps, int mode);
void fir_free(FirState &s);
// Empty history; the device writes are queued on `stream` (ordered with the blocks in flight there).
void fir_reset(FirState &s, hipStream_t stream);
// ev_begin/ev_end (optional) are recorded around the compute kernel(s).  mixpart (optional): [ceil(N/32)][nframes] floats
// that receive, per 32-channel tile and frame, the sum of the block's outputs (the Output node's mix bus, first stage).
int fir_process(FirState &s, const float *in, float *out, uint32_t nframes, int hop, float hop_div,
                const Layout &lay, hipStream_t stream, hipEvent_t ev_begin = nullptr, hipEvent_t ev_end = nullptr,
                float *mixpart = nullptr, const FirMixPipe *mixpipe = nullptr);
// Exported state: a FIR_STATE_HEADER-byte header (n_seen, deque length, VecDeque capacity / head, tap count) + the deque's
// samples [held][N] f32, oldest first (fir_kernels.hip, fir_state_export).
constexpr size_t FIR_STATE_HEADER = 32;
size_t fir_state_bytes(const FirState &s);
int fir_state_export(FirState &s, void *host_dst);
// bytes the blob at host_src must have (read from its header), or -1 when `size` cannot even hold a header
int64_t fir_state_import_bytes(const FirState &s, const void *host_src, size_t size);
int fir_state_import(FirState &s, const void *host_src);
// Placement tuning runs real blocks through the node.  fir_park snapshots what up to `nframes` more samples overwrite (the
// ring rows they land in, the non-finite flags, the fill phase's running sums, the host-side deque model), fir_rewind
// puts the host-side counters back before each probe run, fir_unpark restores everything.
struct FirPark {
    float *rows = nullptr;
    unsigned long long *nf = nullptr;
    double *acc = nullptr;
    uint32_t nframes = 0;
    uint64_t n_seen = 0, front = 0;
    uint32_t dq_cap = 0, dq_head = 0;
    bool warm_ok = false;
    const char *last_kernel = nullptr;
};
int fir_park(FirState &s, uint32_t nframes, hipStream_t stream, FirPark &p);
void fir_rewind(FirState &s, const FirPark &p);
int fir_unpark(FirState &s, FirPark &p, hipStream_t stream);
void fir_park_free(FirPark &p);
const char *fir_kernel_name(const FirState &s);
const char *fir_last_error();

}  // namespace dspfx
