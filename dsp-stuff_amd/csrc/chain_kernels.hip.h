// chain_kernels.hip.h -- gfx950 device code for the fused effect chain.
//
// One wavefront lane owns CPL adjacent channels; a workgroup of 256 lanes owns
// 256*CPL adjacent channels.  In both sample layouts (frame-major [frame][channel] and
// channel-tiled [channel tile][frame][channel in tile], see Layout) the channels of one frame
// are contiguous, so every load/store a wave issues is one coalesced burst of 64*CPL floats.
// The time axis is walked in chunks of F frames held in registers; every node of
// the chain is applied to the chunk before the next chunk is touched, so samples
// make exactly one HBM round trip per block (16 B/sample with one delay line).
// Filter state lives in registers for the whole block (loaded/stored once).
//
// Arithmetic follows the reference literally (file:line cited per function,
// relative to /root/reference/): f32, left-to-right, NO fused multiply-add
// (the translation unit is compiled with -ffp-contract=off), IEEE division.
#pragma once
#ifndef __HIPCC_RTC__            // hiprtc (run-time specialisation, see jit.hip) brings its own runtime declarations
#include <hip/hip_runtime.h>
#include <stdint.h>
#endif

namespace dspfx {

constexpr int MAX_SLOTS = 8;   // nodes fused into one kernel launch
constexpr int WG = 256;        // workgroup size (4 waves)

// kinds: numeric values == dspfx_kind in include/dspfx.h
enum : int {
    K_GAIN = 0, K_BIQUAD = 1, K_LOW_PASS = 2, K_HIGH_PASS = 3, K_REVERB = 4, K_DISTORT = 5,
    K_OVERDRIVE = 6, K_CHEBYSHEV = 7, K_FIR = 8, K_ADD = 9, K_MIX = 10, K_SIGNAL_GEN = 11, K_ENVELOPE = 12
};
enum : int { G_SINE = 0, G_TRIANGLE = 1, G_SQUARE = 2, G_CONSTANT = 3 };   // signal_gen.rs:17-22
// Interpreter-only pseudo mode: Square and Constant share one code path, told apart by a per-sample select
// instead of a branch.  (A separate wave-uniform branch for Constant was dropped by the gfx950 backend in the
// guarded tail instantiation -- the optimised IR still had it, the ISA did not; caught by the ragged-N test.)
constexpr int G_SQUARE_OR_CONST = 4;
enum : int {
    D_HARD_CLIP = 0, D_SOFT_CLIP = 1, D_TANH = 2, D_RECIP_SOFT_CLIP = 3, D_FUZZ = 4, D_SIN = 5,
    D_ATAN = 6, D_SQUARE = 7, D_CHEBYSHEV4 = 8
};

// Per-node kernel arguments (wave-uniform: they live in SGPRs).
//   BIQUAD : p = {a1,a2,b0,b1,b2} already divided by a0 on the host (biquad.rs:62-76)
//   REVERB : p[0] = decay; groups = ring group table; pos = ring row of the block's first frame
//   others : p = the reference's slider values in field order
struct SlotArgs {
    int kind;
    int mode;
    float p[6];
    float *state;
    float *const *groups;   // REVERB: device table of ring group base pointers
    // ... and, for launches of at most 128 frames, the three groups the block's rows can lie in, straight from the host (which
    // keeps the table): the group of row `pos`, the one after it, group 0 (rows past the wrap at D).  The time-sliced kernel
    // (few channels: nothing hides a wave's latency) forms its tap addresses from these and saves the dependent scalar loads of
    // the table before its first tap load; the other kernels read the table.  g_ia = index of g_a; g_valid = 0: not filled.
    float *g_a, *g_b, *g_0;
    unsigned g_ia;
    int g_valid;
    unsigned D;
    unsigned pos;
    int hop;     // apply collect_and_average (one pipe) to this node's input
    // REVERB: the first zero_rows frames of this launch read their taps as +0.0 whatever the ring rows hold.  This is how a
    // NEW zero-filled ring (Reverb::refresh_seconds, reverb.rs:55-71 -- run by the reference on every slider change of the
    // node, dsp-stuff-derive/src/lib.rs:560-568 -- and dspfx_reset) costs nothing: the host counts D frames down from the
    // clear instead of rewriting up to 94 GiB of rows; the launches keep overwriting the rows they read, so after D frames the
    // ring holds only samples written since.  0 in steady state (one scalar compare per chunk).
    unsigned zero_rows;
    double rc;   // DISTORT Hard/SoftClip: f64 1/level for the exact fast division (see div_c)
    // control ports (`as_input` sliders, dsp-stuff-derive/src/lib.rs:122-161), slider field order:
    const float *ctl[3];   // connected port: signal in the sample layout, else nullptr
    float *latch[3];       // per-channel latched slider value [N] (lib.rs:148), or nullptr
    int latch_valid;       // bit k: slider k currently holds per-channel latched values
    int pad2_;
};

struct ChainArgs {
    const float *in;
    const float *side;   // port "b" of ADD/MIX, or nullptr
    float *out;
    float *mixpart;      // [mix_stride rows][nframes] partial sums of the mix bus, one row per WORKGROUP (contiguous), or nullptr
    unsigned N;
    unsigned nframes;
    float hop_div;       // f32(0.0001 + 1.0)  (node.rs:166,179)
    int n_slots;
    int side_hop;
    unsigned mix_stride; // rows of mixpart = number of workgroups over all launches of this block
    unsigned c_base;     // first channel of this launch
    unsigned n_launch;   // channels covered by this launch
    unsigned wave_base;  // mixpart row of this launch's first workgroup
    // sample layout: channel c, frame f lives at  (c >> w_shift) * tile_stride + f * ld + (c & w_mask)
    //   frame-major [B][N]      : w_shift = 31, w_mask = ~0u>>1, ld = N, tile strides unused
    //   channel-tiled [N/W][B][W]: w_shift = log2 W, w_mask = W-1, ld = W, tile_stride = B*W
    unsigned w_shift;
    unsigned w_mask;
    unsigned ld;
    unsigned pad_;
    size_t io_tile_stride;   // floats between consecutive channel tiles of in/out/side
    double hop_rc;           // f64 1/hop_div
    double third_rc;         // f64 1/3.0f  (SoftClip's powi(3)/3.0)
    int fast_div;            // every constant divisor of this launch passed the exhaustive check
    int xcd_remap;           // 1: blocks of one XCD (b % 8) cover a contiguous range of channel tiles
    // Pipelined mix bus (mixpipe_prologue): the second and third reduction stage of EARLIER blocks ride in this
    // launch's first workgroups instead of separate kernels on a second stream.
    int skip_store;          // in-place launch of an empty chain that only feeds the mix bus: nothing to write back
    int mp_stage;            // bit 0: slice-reduce mp_prev_a into mp_cur_b; bit 1: final-reduce mp_prev_b into mp_mix
    unsigned mp_rows_a;      // rows (waves) of mp_prev_a
    const float *mp_prev_a;  // per-wave partials [rows][nframes] of the previous block
    float *mp_cur_b;         // [MIX_SLICES][nframes] slice sums of the previous block (written here)
    const float *mp_prev_b;  // slice sums of the block before that (written by the previous launch)
    float *mp_mix;           // [nframes] mix bus of the block two launches back
    float mp_div;            // Output-node divisor f32(0.0001 + n), or 0 to leave the sums un-normalised
    // Same-block mix bus (mix_tail): the slice and final stages of THIS block run inside this launch -- the workgroup
    // that completes a slice of rows reduces it, the one that completes the last slice finishes the bus.
    float mt_div;            // Output-node divisor, or 0
    // 1: every WAVE of a 256-lane workgroup leaves its own row of partial sums (row = wave_base + 4 workgroup + wave) instead
    // of one row per workgroup.  Engines below 16384 channels run this way, so that their rows are the time-sliced kernels'
    // rows -- one per 64 channels -- and the bus' summation order does not change when such an engine moves from the
    // interpreter to the kernels the background compiler made for it (jit.hip), whenever that happens.
    int mix_per_wave;
    unsigned *mt_tickets;    // [MIX_SLICES + 1] arrival counters, zero between launches; nullptr: no tail
    float *mt_part2;         // [MIX_SLICES][nframes] slice sums
    float *mt_mix;           // [nframes] the bus of this block
    SlotArgs slot[MAX_SLOTS];
};

// ---- mix bus, second and third stage (shared by the stand-alone kernels and the in-kernel pipeline) --------
// part is [rows][nframes] (one contiguous row per workgroup of a chain kernel / per 32-channel tile of a FIR sweep).  Slice
// stage: slice b sums a fixed range of rows for every frame (lane = frame: coalesced row reads) into part2[b][frame]; final
// stage: the slices are summed in fixed order.  Fixed association => run-to-run deterministic, and identical in every form
// (stand-alone kernels, pipelined in later launches, in the tail of the same launch).
#ifndef DSPFX_MIX_SLICES
#define DSPFX_MIX_SLICES 64
#endif
constexpr unsigned MIX_SLICES = DSPFX_MIX_SLICES;   // <= 128 (the engine's slice buffers)
// Rows per slice: at least 32 (one batch of the in-launch tail's loads), so a launch of few workgroups gets few slices and its
// final stage is one batch too; 64 slices from 2048 rows on.  Every form of the bus cuts its rows the same way.
__host__ __device__ __forceinline__ unsigned mix_rows_per_slice(unsigned rows) {
    const unsigned per = (rows + MIX_SLICES - 1) / MIX_SLICES;
    return per < 32u ? 32u : per;
}
__device__ __forceinline__ float mix_rows_sum(const float *src, unsigned n, unsigned nframes, unsigned f) {
    float a0 = 0.0f, a1 = 0.0f;
    unsigned r = 0;
    for (; r + 1 < n; r += 2) {
        const float x0 = src[(size_t)r * nframes + f], x1 = src[(size_t)(r + 1) * nframes + f];
        a0 = a0 + x0;
        a1 = a1 + x1;
    }
    if (r < n) a0 = a0 + src[(size_t)r * nframes + f];
    return a0 + a1;
}
__device__ __forceinline__ void mix_slice_reduce(const float *part, float *part2, unsigned waves, unsigned nframes,
                                                 unsigned b, unsigned tid, unsigned nthreads) {
    const unsigned per = mix_rows_per_slice(waves);
    const unsigned w0 = min(waves, b * per), w1 = min(waves, w0 + per);
    for (unsigned f = tid; f < nframes; f += nthreads) part2[(size_t)b * nframes + f] = mix_rows_sum(part + (size_t)w0 * nframes, w1 - w0, nframes, f);
}
// (slices that hold no rows were written as +0 by the slice stage and are added like the others)
__device__ __forceinline__ void mix_final_reduce(const float *part2, float *mix, unsigned nframes, float div,
                                                 unsigned tid, unsigned nthreads) {
    for (unsigned f = tid; f < nframes; f += nthreads) {
        const float acc = mix_rows_sum(part2, MIX_SLICES, nframes, f);
        mix[f] = div != 0.0f ? acc / div : acc;     // node.rs:189-191 (the Output node's hop)
    }
}
// The first MIX_SLICES workgroups of a launch reduce one slice each of the previous block's partials, the next one
// finishes the block before that.  Both only read what EARLIER launches on the same stream wrote.
__device__ __forceinline__ void mixpipe_prologue(const ChainArgs &a) {
    const unsigned b = blockIdx.x;
    if ((a.mp_stage & 1) && b < MIX_SLICES) mix_slice_reduce(a.mp_prev_a, a.mp_cur_b, a.mp_rows_a, a.nframes, b, threadIdx.x, blockDim.x);
    else if ((a.mp_stage & 2) && b == MIX_SLICES) mix_final_reduce(a.mp_prev_b, a.mp_mix, a.nframes, a.mp_div, threadIdx.x, blockDim.x);
}

// The kernel's arguments as they lie in the kernarg segment (every chain / graph kernel takes its ChainArgs first), for code
// that runs once or twice per launch: read there, at the point of use, the fields it needs do not sit in scalar registers
// across the chunk loop (the by-value argument's loads are loop-invariant and get hoisted: with the bus' tail fields the
// 5-node kernel went from 18 to 154 scalar spills and lost a wave of occupancy to the spill registers).
struct ColdArgs {
    float *mixpart;
    unsigned nframes, mix_stride, wave_base, n_launch;
    int xcd_remap;
    float mt_div;
    int mix_per_wave;
    unsigned *mt_tickets;
    float *mt_part2, *mt_mix;
};
__device__ __forceinline__ ColdArgs cold_args() {
    typedef const __attribute__((address_space(4))) ChainArgs *KernArgPtr;      // constant address space: scalar loads, wave-uniform values
    KernArgPtr ka = (KernArgPtr)__builtin_amdgcn_kernarg_segment_ptr();
    asm volatile("" : "+s"(ka));               // opaque: nothing read through it can be hoisted above this point
    ColdArgs c;
    c.mixpart = ka->mixpart;
    c.nframes = ka->nframes;
    c.mix_stride = ka->mix_stride;
    c.wave_base = ka->wave_base;
    c.n_launch = ka->n_launch;
    c.xcd_remap = ka->xcd_remap;
    c.mt_div = ka->mt_div;
    c.mix_per_wave = ka->mix_per_wave;
    c.mt_tickets = ka->mt_tickets;
    c.mt_part2 = ka->mt_part2;
    c.mt_mix = ka->mt_mix;
    return c;
}
// ---- same-block mix bus: the slice and final stages in the tail of the launch that produced the rows --------------
// Rows are handed between workgroups INSIDE a launch, across XCDs whose L2s are not coherent with each other: the rows
// and slice sums are written through (sc1 stores), every writing wave drains its stores (s_waitcnt vmcnt(0)) before
// ONE lane bumps an agent-scope counter, and the workgroup that draws the last ticket reads them back with sc1 loads
// (which the caches cannot serve stale).  No fences: a release fence would write back the XCD's whole L2, which at
// this point is full of freshly stored samples.  Counters are reset by the last arriver, so they are zero between launches.
// DSPFX_BUS_FENCE (make libdspfx_busfence.so): the same hand-over inside the HIP memory model -- plain stores, an agent-scope
// RELEASE fence before a row's ticket, an agent-scope ACQUIRE fence behind the last ticket, plain loads -- for bisecting should a
// compiler or firmware update break the default form (which rests on gfx950's documented write-through / sc1 behaviour, not on
// the model).  Bit-identical results; slower: the release writes back the XCD's whole L2, full of freshly stored samples
// (A/B table: profiles/r04_bus_fence_ab.txt; tests/test_gpu_threads.py runs both forms against each other).
typedef __attribute__((address_space(1))) unsigned dspfx_gu32;   // every shared word is a GLOBAL agent-scope access, never flat
__device__ __forceinline__ float ld_sc1(const float *p) {
    return __uint_as_float(__hip_atomic_load((const dspfx_gu32 *)reinterpret_cast<const unsigned *>(p), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
}
__device__ __forceinline__ void st_sc1(float *p, float v) {
#ifdef DSPFX_BUS_FENCE
    *p = v;                                           // published by the release fence in bus_publish()
#else
    __hip_atomic_store((dspfx_gu32 *)reinterpret_cast<unsigned *>(p), __float_as_uint(v), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#endif
}
// the writing wave, after its last store of a row / slice sum and before its ticket
__device__ __forceinline__ void bus_publish() {
#ifdef DSPFX_BUS_FENCE
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
#endif
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // (behind a release fence too: the compiler may drop the fence's own wait)
}
// the wave that drew the last ticket, before it reads what the others published
__device__ __forceinline__ void bus_acquire() {
#ifdef DSPFX_BUS_FENCE
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
#endif
}
__device__ __forceinline__ unsigned ticket_take(unsigned *t, int lane) {
    unsigned v = 0;
    if (lane == 0) v = __hip_atomic_fetch_add((dspfx_gu32 *)t, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    return (unsigned)__builtin_amdgcn_readfirstlane((int)v);
}
// dst[f] = fixed-order sum over rows r of src[r][f], r in [0, n): rows of even index (counted from the first) are
// added one after another into one accumulator, rows of odd index into a second one, the result is their sum -- the
// association of mix_slice_reduce / mix_final_reduce.  ONE wave; the rows are read with sc1 loads, TAIL_BATCH of them in
// flight per lane (a dependent chain of single loads would cost a memory round trip per row: the whole tail is latency).
#ifndef DSPFX_TAIL_BATCH
#define DSPFX_TAIL_BATCH 16    // 16 and 32 rows in flight measure the same (profiles/r03_bus_ab.txt); 16 leaves the small kernels their occupancy
#endif
constexpr int TAIL_BATCH = DSPFX_TAIL_BATCH;   // even
typedef unsigned dspfx_u32x2 __attribute__((ext_vector_type(2)));
typedef unsigned dspfx_u32x4 __attribute__((ext_vector_type(4)));
// sc1 loads through a buffer descriptor (`buffer_load_dwordx2 v, v_off, s[rsrc], 0 offen sc1`): one running 32-bit offset
// register instead of a 64-bit vector address per row (the atomic-load builtin's global sc1 loads: 64 VGPRs of addresses for
// 32 rows in flight), and the descriptor's bounds check returns 0 for rows past n_valid -- no branches in the batch.
// nf is even (the host takes this path for even block lengths only): a lane takes the frames 2 lane, 2 lane + 1.
#ifdef DSPFX_BUS_FENCE
constexpr int BUF_AUX_SC1 = 0;                        // plain loads behind the acquire fence
#else
constexpr int BUF_AUX_SC1 = 16;
#endif
__device__ __forceinline__ void tail_reduce_rows(const float *src, unsigned n, unsigned n_valid, unsigned nf, float *dst, float div, bool write_through, int lane) {
    const unsigned stride = nf * (unsigned)sizeof(float);           // n * stride stays far below 4 GiB (<= 8192 rows of <= a few thousand frames)
    const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(src), 0, (int)(n_valid * stride), 0x00020000);
    for (unsigned f = (unsigned)lane * 2; f < nf; f += 128) {
        float a00 = 0.0f, a01 = 0.0f, a10 = 0.0f, a11 = 0.0f;       // a<parity of the row><frame>
#pragma unroll 1                                                     // one batch in flight: two would double the registers of the whole kernel
        for (unsigned r0 = 0; r0 < n; r0 += TAIL_BATCH) {
            dspfx_u32x2 x[TAIL_BATCH];
            unsigned off = r0 * stride + f * (unsigned)sizeof(float);
#pragma unroll
            for (int k = 0; k < TAIL_BATCH; ++k, off += stride) x[k] = __builtin_amdgcn_raw_buffer_load_b64(rsrc, off, 0, BUF_AUX_SC1);
#pragma unroll
            for (int k = 0; k < TAIL_BATCH; k += 2) {               // TAIL_BATCH is even: the parity of k is the row's
                const bool in0 = r0 + k < n, in1 = r0 + k + 1 < n;  // uniform
                const float s00 = a00 + __uint_as_float(x[k].x), s01 = a01 + __uint_as_float(x[k].y);
                const float s10 = a10 + __uint_as_float(x[k + 1].x), s11 = a11 + __uint_as_float(x[k + 1].y);
                a00 = in0 ? s00 : a00;
                a01 = in0 ? s01 : a01;
                a10 = in1 ? s10 : a10;
                a11 = in1 ? s11 : a11;
            }
        }
        float t0 = a00 + a10, t1 = a01 + a11;
        if (div != 0.0f) {                                          // node.rs:189-191 (the Output node's hop)
            t0 = t0 / div;
            t1 = t1 / div;
        }
        if (write_through) {
            st_sc1(dst + f, t0);
            st_sc1(dst + f + 1, t1);
        } else {
            dst[f] = t0;
            dst[f + 1] = t1;
        }
    }
}
// Called by ONE wave per row (the wave that stored the row), all 64 lanes alive, after the row's last store.
__device__ __forceinline__ void mix_tail(unsigned row, int lane) {
    bus_publish();                                                    // the row has left this wave and is in memory
    const ColdArgs a = cold_args();
    const unsigned rows = a.mix_stride, nf = a.nframes;
    const unsigned per = mix_rows_per_slice(rows);
    const unsigned b = row / per, w0 = b * per, w1 = min(rows, w0 + per);
    if (ticket_take(a.mt_tickets + b, lane) + 1 != w1 - w0) return;
    bus_acquire();
    // last row of slice b: the same sums in the same order as mix_slice_reduce
    tail_reduce_rows(a.mixpart + (size_t)w0 * nf, w1 - w0, w1 - w0, nf, a.mt_part2 + (size_t)b * nf, 0.0f, true, lane);
    if (lane == 0) __hip_atomic_store((dspfx_gu32 *)(a.mt_tickets + b), 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    bus_publish();
    const unsigned live = (rows + per - 1) / per;                     // slices that hold rows; the others count as the +0 their stage would have written
    if (ticket_take(a.mt_tickets + MIX_SLICES, lane) + 1 != live) return;
    bus_acquire();
    // last slice: the same sums in the same order as mix_final_reduce
    tail_reduce_rows(a.mt_part2, MIX_SLICES, live, nf, a.mt_mix, a.mt_div, false, lane);      // (batches past `live` load nothing: bounds check)
    if (lane == 0) __hip_atomic_store((dspfx_gu32 *)(a.mt_tickets + MIX_SLICES), 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// Sample layout shared by every kernel: channel c, frame f lives at
//   (c >> w_shift) * tile_stride + f * ld + (c & w_mask)
//   frame-major [B][N]       : w_shift = 31, w_mask = 0x7fffffff, ld = N, tile_stride = 0
//   channel-tiled [N/W][B][W]: w_shift = log2 W, w_mask = W-1, ld = W, tile_stride = B*W
// Delay rings are stored as separately allocated GROUPS of 128 rows behind a pointer table:
// ring row r (0..D-1) of channel tile t, channel-in-tile cw, lives at
//   groups[r >> 7] + (t * 128 + (r & 127)) * ld + cw            ([ceil(D/128)] x [ntiles][128][W])
// so the 128 rows a block reads/overwrites are, for ALL tiles together, one or two contiguous
// N*128*4-byte extents (TLB / DRAM-page friendly at a 94 GiB footprint), and every group can be placed
// -- and re-placed -- independently: some physical HBM regions stream ~18 % slower for this pattern
// (profiles/r01_placement.txt), the engine probes each group at setup and re-allocates the slow ones.
// With a single tile (frame-major, W = N) a group is the plain [128][N].
constexpr unsigned RING_GROUP_ROWS = 128;
__host__ __device__ inline size_t ring_in_group_offset(unsigned r, size_t tile, size_t ld) {
    return (tile * RING_GROUP_ROWS + (r & 127u)) * ld;
}

struct Layout {
    unsigned w_shift;
    unsigned w_mask;
    unsigned ld;
    unsigned pad_;
    size_t tile_stride;
    __host__ __device__ size_t at(unsigned f, size_t c) const {
        return (c >> w_shift) * tile_stride + (size_t)f * ld + (c & w_mask);
    }
};

// ---- static signature encoding ------------------------------------------------
// A slot signature is either SIG_DYN (kind/mode/hop read from SlotArgs at run time,
// wave-uniform switch) or a compile-time triple packed by sig().
constexpr int SIG_DYN = -1;
constexpr int SIG_NONE = -2;
constexpr int sig(int kind, int mode = 0, int hop = 0) { return kind | (mode << 8) | (hop << 16); }
constexpr int sig_kind(int s) { return s & 0xff; }
constexpr int sig_mode(int s) { return (s >> 8) & 0xff; }
constexpr int sig_hop(int s) { return (s >> 16) & 1; }
template <int KIND> constexpr bool sig_is(int s) { return s >= 0 && sig_kind(s) == KIND; }

// Sample/ring traffic is streamed exactly once per block (reuse distance = a whole
// delay period), so those loads/stores carry the nontemporal hint: measured 1.13-1.27x
// on the 5-node chain (interleaved A/B, profiles/r01_ab_nt.txt).  DSPFX_NT is a mask over
// the four streams for A/B builds; 15 = all.
#ifndef DSPFX_NT
#define DSPFX_NT 15
#endif
template <int CPL> struct NVecT;
template <> struct NVecT<1> { using type = float; };
template <> struct NVecT<2> { typedef float type __attribute__((ext_vector_type(2))); };
template <> struct NVecT<4> { typedef float type __attribute__((ext_vector_type(4))); };

// streams (DSPFX_NT is a mask over them)
constexpr int S_IN = 1, S_RING_LD = 2, S_RING_ST = 4, S_OUT = 8, S_STATE = 0;
constexpr bool nt_for(int stream) { return (DSPFX_NT & stream) != 0; }

// Every pointer the kernels dereference is hipMalloc'ed device memory: the explicit global address space
// turns loads through pointers that were themselves loaded (delay-ring group table) from FLAT into GLOBAL
// instructions.  FLAT operations also count on lgkmcnt, so the scalar loads of the next group pointer used
// to wait for the previous frame's ring load (measured on the ISA: serialised memory latencies per chunk).
#define DSPFX_GLOBAL __attribute__((address_space(1)))
template <int CPL, bool NT>
__device__ __forceinline__ void load_vec_raw(const float *p, float (&v)[CPL]) {
    using V = typename NVecT<CPL>::type;
    const DSPFX_GLOBAL V *gp = (const DSPFX_GLOBAL V *)reinterpret_cast<const V *>(p);
    V t;
    if constexpr (NT) t = __builtin_nontemporal_load(gp);
    else t = *gp;
    if constexpr (CPL == 1) { v[0] = t; }
    else {
#pragma unroll
        for (int j = 0; j < CPL; ++j) v[j] = t[j];
    }
}
template <int CPL, bool NT>
__device__ __forceinline__ void store_vec_raw(float *p, const float (&v)[CPL]) {
    using V = typename NVecT<CPL>::type;
    V t;
    if constexpr (CPL == 1) { t = v[0]; }
    else {
#pragma unroll
        for (int j = 0; j < CPL; ++j) t[j] = v[j];
    }
    DSPFX_GLOBAL V *gp = (DSPFX_GLOBAL V *)reinterpret_cast<V *>(p);
    if constexpr (NT) __builtin_nontemporal_store(t, gp);
    else *gp = t;
}

// GUARD=true is the one-wave tail launch for N % (64*CPL) != 0: out-of-range lanes
// stay alive (the mix-bus reduction shuffles across the wave), read zeros and
// store nothing.  GUARD=false launches cover whole waves only.
template <int CPL, bool GUARD, int STREAM = S_STATE>
__device__ __forceinline__ void load_vec(const float *p, float (&v)[CPL], bool active) {
    if constexpr (GUARD) {
        if (active) load_vec_raw<CPL, nt_for(STREAM)>(p, v);
        else {
#pragma unroll
            for (int j = 0; j < CPL; ++j) v[j] = 0.0f;
        }
    } else {
        load_vec_raw<CPL, nt_for(STREAM)>(p, v);
    }
}
template <int CPL, bool GUARD, int STREAM = S_STATE>
__device__ __forceinline__ void store_vec(float *p, const float (&v)[CPL], bool active) {
    if constexpr (GUARD) {
        if (active) store_vec_raw<CPL, nt_for(STREAM)>(p, v);
    } else {
        store_vec_raw<CPL, nt_for(STREAM)>(p, v);
    }
}

// ---- reference arithmetic, one sample ------------------------------------------

// x / c for a wave-uniform constant c.  FAST: (float)((double)x * rc) with rc = RN_f64(1/c):
// the exact quotient of two f32 values is never closer than 2^-49 (relative) to an f32
// rounding boundary while the f64 product is within 2^-52 of it, so the single final
// rounding equals IEEE f32 division.  Subnormal quotients have fewer than 24 bits, so there an exact quotient CAN be
// a tie: x / c = (2k+1) 2^-150 needs x = (c/2)(2k+1) 2^-149 on the subnormal grid, i.e. c/2 an integer -- ties exist
// only for EVEN INTEGER c (and any quotient that is not a tie stays >= 2^-48 away from one).  The host therefore
// takes FAST for every other constant outright and runs the exhaustive 2^32-input check (verify_div_kernel) for
// even integers that are not powers of two; those that fail it take the IEEE path (FAST = false).
template <bool FAST>
__device__ __forceinline__ float div_c(float x, float c, double rc) {
    if constexpr (FAST) return (float)((double)x * rc);
    else return x / c;
}

// x / c for a per-lane divisor c with rc = 1.0 / (double)c: the f64 product rounds to the IEEE f32 quotient whenever
// that quotient is a normal number (no f32 quotient lies within 2^-49 of a rounding boundary unless it is an exact
// tie, and ties only exist among subnormal results); subnormal results take the IEEE division.
__device__ __forceinline__ float div_lane(float x, float c, double rc) {
    float q = (float)((double)x * rc);
    if (!(__builtin_fabsf(q) >= 0x1p-126f)) q = x / c;
    return q;
}

// node.rs:162-194 with one connected pipe: buf = 0.0; buf += x; buf /= 0.0001f + 1.0f
template <bool FAST>
__device__ __forceinline__ float link_hop(float x, float div, double rc) { return div_c<FAST>(0.0f + x, div, rc); }

// distort.rs:53-61
__device__ __forceinline__ float clip1(float s) { return s < -1.0f ? -1.0f : (s > 1.0f ? 1.0f : s); }
// f32::signum: +-1 by sign bit, NaN stays NaN
__device__ __forceinline__ float rs_signum(float x) {
    return x != x ? x : (__float_as_uint(x) >> 31 ? -1.0f : 1.0f);
}

// libm-backed modes: the reference calls glibc's tanhf/sinf/atanf/expf through Rust std.
// Evaluating in f64 and rounding once gives the correctly rounded f32 result, which is
// within 1 ulp of glibc's sinf/atanf/expf and 2 ulp of its tanhf (measured over all
// arguments in range) -- closer than ocml's f32 routines.
// The library's f64 tanh / sin are accurate to an f64 ulp and cost accordingly (a tanh-only chain ran at 1.7 TB/s).
// An f32 result needs far less: the f64 evaluations below are good to ~1e-15 relative, so after the single rounding
// to f32 they equal the library path except on a few near-tie inputs out of 2^32 (dspfx_verify_libm counts them,
// exhaustively, and the largest difference is 1 ulp) -- the same distance from glibc as before.
__device__ __forceinline__ float tanh_lib(float x) { return (float)tanh((double)x); }
__device__ __forceinline__ float sin_lib(float x) { return (float)sin((double)x); }

// e^y for -160 <= y <= 40: y = k ln2 + r, |r| <= ln2/2, degree-12 Taylor (r^13/13! < 2e-16), scaled by 2^k
__device__ __forceinline__ double exp_f64(double y) {
    const double k = __builtin_rint(y * 1.44269504088896338700e+00);
    double r = __builtin_fma(-k, 6.93147180369123816490e-01, y);      // ln2 high part (fdlibm split)
    r = __builtin_fma(-k, 1.90821492927058770002e-10, r);               // ln2 low part
    double p = 1.0 / 479001600.0;
    p = __builtin_fma(p, r, 1.0 / 39916800.0);
    p = __builtin_fma(p, r, 1.0 / 3628800.0);
    p = __builtin_fma(p, r, 1.0 / 362880.0);
    p = __builtin_fma(p, r, 1.0 / 40320.0);
    p = __builtin_fma(p, r, 1.0 / 5040.0);
    p = __builtin_fma(p, r, 1.0 / 720.0);
    p = __builtin_fma(p, r, 1.0 / 120.0);
    p = __builtin_fma(p, r, 1.0 / 24.0);
    p = __builtin_fma(p, r, 1.0 / 6.0);
    p = __builtin_fma(p, r, 0.5);
    p = __builtin_fma(p, r, 1.0);
    p = __builtin_fma(p, r, 1.0);
    return __builtin_ldexp(p, (int)k);
}
// tanh|x| = 1 - 2 / (e^(2|x|) + 1); below 2^-7 the odd series (the difference would cancel); |x| >= 20 saturates
__device__ __forceinline__ float tanh_cr(float x) {
    const double ax = __builtin_fabs((double)x);
    const double y = __builtin_fmin(ax + ax, 40.0);
    const double d = exp_f64(y) + 1.0;
    double rc = __builtin_amdgcn_rcp(d);
    rc = __builtin_fma(__builtin_fma(-d, rc, 1.0), rc, rc);           // two Newton steps: full f64 accuracy
    rc = __builtin_fma(__builtin_fma(-d, rc, 1.0), rc, rc);
    const double big = __builtin_fma(-2.0, rc, 1.0);
    const double x2 = ax * ax;
    const double small = ax * __builtin_fma(x2, __builtin_fma(x2, __builtin_fma(x2, -17.0 / 315.0, 2.0 / 15.0), -1.0 / 3.0), 1.0);
    const float t = (float)(ax < 0x1p-7 ? small : big);
    return x != x ? x : __builtin_copysignf(t, x);
}
// sin x = +-sin r or +-cos r with x = n pi/2 + r, |r| <= pi/4 (two-part pi/2, fused), fdlibm's kernel polynomials;
// |x| >= 2^22, infinities and NaN take the library routine (a divergent branch nobody takes on audio)
__device__ __forceinline__ float sin_cr(float x) {
    if (!(__builtin_fabsf(x) < 0x1p22f)) return sin_lib(x);
    const double xd = (double)x;
    const double n = __builtin_rint(xd * 6.36619772367581382433e-01);
    double r = __builtin_fma(-n, 1.57079632679489655800e+00, xd);
    r = __builtin_fma(-n, 6.12323399573676603587e-17, r);
    const double z = r * r;
    double sp = 1.58969099521155010221e-10;
    sp = __builtin_fma(sp, z, -2.50507602534068634195e-08);
    sp = __builtin_fma(sp, z, 2.75573137070700676789e-06);
    sp = __builtin_fma(sp, z, -1.98412698298579493134e-04);
    sp = __builtin_fma(sp, z, 8.33333333332248946124e-03);
    sp = __builtin_fma(sp, z, -1.66666666666666324348e-01);
    const double sv = __builtin_fma(r * z, sp, r);
    double cp = -1.13596475577881948265e-11;
    cp = __builtin_fma(cp, z, 2.08757232129817482790e-09);
    cp = __builtin_fma(cp, z, -2.75573143513906633035e-07);
    cp = __builtin_fma(cp, z, 2.48015872894767294178e-05);
    cp = __builtin_fma(cp, z, -1.38888888888741095749e-03);
    cp = __builtin_fma(cp, z, 4.16666666666666019037e-02);
    const double cv = __builtin_fma(z * z, cp, __builtin_fma(z, -0.5, 1.0));
    const int q = (int)n;
    const double m = (q & 1) ? cv : sv;
    const float out = (float)((q & 2) ? -m : m);
    return x == 0.0f ? x : out;                   // keeps -0 (the reduction turns it into +0)
}
__device__ __forceinline__ float atan_lib(float x) { return (float)atan((double)x); }
// atan|x| = base + atan(w), |w| <= tan(pi/8), with ONE division:  |x| <= tan(pi/8): w = |x|;
// |x| <= 1/tan(pi/8): w = (|x|-1)/(|x|+1), base pi/4;  beyond: w = -1/|x|, base pi/2.
// atan(w) = w + w^3 P(w^2), P = degree-11 Chebyshev interpolant of the series (rel. error 3e-18 in exact arithmetic)
__device__ __forceinline__ float atan_cr(float x) {
    const double a = __builtin_fmin(__builtin_fabs((double)x), 1e40);
    const bool lo = a <= 4.14213562373095034e-01, hi = a > 2.41421356237309492e+00;
    const double num = lo ? a : (hi ? -1.0 : a - 1.0);
    const double den = lo ? 1.0 : (hi ? a : a + 1.0);
    double rc = __builtin_amdgcn_rcp(den);
    rc = __builtin_fma(__builtin_fma(-den, rc, 1.0), rc, rc);
    rc = __builtin_fma(__builtin_fma(-den, rc, 1.0), rc, rc);
    double w = num * rc;
    w = __builtin_fma(__builtin_fma(-w, den, num), rc, w);             // correctly rounded quotient up to an ulp
    const double u = w * w;
    double p = 1.62857568552210278667e-02;
    p = __builtin_fma(p, u, -3.45705619814277442803e-02);
    p = __builtin_fma(p, u, 4.55159322062654927987e-02);
    p = __builtin_fma(p, u, -5.23045427065024423618e-02);
    p = __builtin_fma(p, u, 5.87892899783477515530e-02);
    p = __builtin_fma(p, u, -6.66642488573825492404e-02);
    p = __builtin_fma(p, u, 7.69229637503214269678e-02);
    p = __builtin_fma(p, u, -9.09090875350087729290e-02);
    p = __builtin_fma(p, u, 1.11111111051554467544e-01);
    p = __builtin_fma(p, u, -1.42857142856598284819e-01);
    p = __builtin_fma(p, u, 1.99999999999998040456e-01);
    p = __builtin_fma(p, u, -3.33333333333333314830e-01);
    const double base = lo ? 0.0 : (hi ? 1.57079632679489655800e+00 : 7.85398163397448279000e-01);
    const float t = (float)(base + __builtin_fma(w * u, p, w));
    return x != x ? x : __builtin_copysignf(t, x);
}
__device__ __forceinline__ float exp_lib(float x) { return (float)exp((double)x); }
// e^x rounded once from f64; beyond +-160 / 89 the f32 result is 0 / inf anyway
__device__ __forceinline__ float exp_cr(float x) {
    const double y = __builtin_fmax(__builtin_fmin((double)x, 89.0), -160.0);
    const float r = (float)exp_f64(y);
    return x != x ? x : r;
}
// distort.rs:63-145, every mode except Fuzz, for level >= 0.001 (the `level < 0.001`
// bypass is wave-uniform while level is a slider value and is tested once per chunk
// by the caller).  Branch-free selects: lanes never diverge.
template <int MODE, bool FAST>
__device__ __forceinline__ float distort1(float sample, float level, double level_rc, double third_rc) {
    if constexpr (MODE == D_HARD_CLIP) {          // 63-69
        return div_c<FAST>(clip1(sample * level), level, level_rc);
    } else if constexpr (MODE == D_SOFT_CLIP) {   // 71-86
        const float s = sample * level;
        const float mid = s - div_c<FAST>((s * s) * s, 3.0f, third_rc);   // powi(3) = (s*s)*s
        const float in_range = (s >= -1.0f && s <= 1.0f) ? mid : -2.0f / 3.0f;   // NaN -> else arm
        const float r = s > 1.0f ? 2.0f / 3.0f : in_range;
        return div_c<FAST>(clip1(r), level, level_rc);
    } else if constexpr (MODE == D_TANH) {        // 104-110
        return tanh_cr(sample * level);
    } else if constexpr (MODE == D_RECIP_SOFT_CLIP) {   // 96-102
        return rs_signum(sample) * (1.0f - 1.0f / (fabsf(sample) * level + 1.0f));
    } else if constexpr (MODE == D_SIN) {         // 112-118
        return sin_cr(sample * level);
    } else if constexpr (MODE == D_ATAN) {        // 120-126
        return atan_cr(sample * level);
    } else if constexpr (MODE == D_SQUARE) {      // 128-134
        float v = sample * level;
        return (v * v) * rs_signum(v);
    } else {                                      // D_CHEBYSHEV4, 136-144
        float v = sample * level;
        float v2 = v * v;
        return 8.0f * (v2 * v2) - 8.0f * v2 + 1.0f;
    }
}

// overdrive.rs:31-43, for level >= 0.001 (bypass tested once per chunk by the caller)
__device__ __forceinline__ float overdrive1(float sample, float boost, float drive, float level) {
    const float FRAC_PI_4 = 0.785398163397448309615660845819875721f;
    const float FRAC_2_PI = 0.636619772367581343075535053490057448f;
    float a = sample * boost;
    float b = FRAC_PI_4 * a;
    float c = atan_cr(b);
    float d = FRAC_2_PI * c;
    float mix = drive * d + (1.0f - drive) * sample;
    return mix * level;
}

// chebyshev.rs:28-42
// chebyshev.rs:28-42.  tp / tn are tanh(level_pos) / tanh(level_neg), wave-uniform: computed once per chunk by the
// caller.  Branch-free: a divergent `if` made every lane pay for both branches and both denominators (4 tanh).
__device__ __forceinline__ float chebyshev1(float sample, float lp, float ln, float tp, float tn) {
    const bool pos = sample >= 0.0f;               // NaN takes the negative branch, like the reference
    const float l = pos ? lp : ln;
    const float r = tanh_cr(sample * l) / (pos ? tp : tn);
    return l < 0.001f ? sample : r;
}

// ---- feedback delay line (reverb.rs:86-103), split so the tap loads can be issued early ---------
// The taps of a chunk do not depend on anything the chunk computes (nframes <= D), so a statically
// specialised kernel issues them together with the sample loads at the top of the chunk instead of after
// the nodes in front of the delay: one exposed memory latency per chunk instead of two, which is what
// bounds launches with few channels (one wave per SIMD, nothing else to switch to).
// (filled by ring_groups / ring_groups_host below)
struct RingGroups {
    float *ga, *gb, *g0;
    unsigned gia;
};
template <int F, int CPL> struct RingPre {
    float tap[F][CPL];
    float *row[F];     // wave-uniform row base (scalar registers): the base of the row's buffer descriptor, the lane adds cx.ring_off
};
// signal_gen.rs:57-104, one sample: `total` is the block-local phase advance, `clock` the phase carried
// between 128-frame blocks.  Square compares `total` (not the phase) with 0.5, like the reference.
template <int MODE>
__device__ __forceinline__ float signal1(float clock, float &total, float frequency, float amplitude) {
    if constexpr (MODE == G_CONSTANT) {
        return amplitude;                             // do_const: copies the amplitude block
    } else {
        const float TAU = 6.28318530717958647692528676655900577f;
        const float step = frequency / 48000.0f;
        total = total + step;
        if constexpr (MODE == G_SINE) return sin_cr((clock + total) * TAU) * amplitude;
        else if constexpr (MODE == G_TRIANGLE) return (2.0f * fmodf(clock + total, 1.0f) - 1.0f) * amplitude;
        else return (total > 0.5f ? 1.0f : -1.0f) * amplitude;   // G_SQUARE, and G_SQUARE_OR_CONST's square half
    }
}

// ---- per-lane chunk context ------------------------------------------------------
struct Ctx {
    size_t c;        // first channel of this lane
    size_t N;
    // Addresses are split into a wave-uniform 64-bit part (scalar registers, scalar arithmetic) and a 32-bit
    // per-lane part that does not depend on the frame, so every sample / ring access is
    // `global_load/store v, v_off, s[base]`: no 64-bit vector address arithmetic, no VGPR pair per frame.
    size_t io_base0;   // offset of (frame 0, the wave's first channel) in in/out/side/control buffers   [uniform]
    unsigned io_off;   // this lane's channel relative to that, in BYTES (see lane_ptr)                    [lane]
    size_t ring_base0; // (tile0 * 128) * ld + cw0: the wave's first channel inside a ring group, row 0   [uniform]
    unsigned ring_off; // this lane relative to that                                                      [lane]
    size_t ld;       // floats between consecutive frames
    unsigned f0;     // first frame of the chunk
    float hop_div;
    double hop_rc;
    double third_rc;
    const float *side;
    int side_hop;      // bit 0: hop on the side input; bit 1: hop on control links (both are "internal" links)
    bool active;     // false only for padding lanes of the guarded tail launch
};

// Per-wave address bases (see Ctx).  `c` is the lane's first channel; every lane of a wave has c >= c0.
struct WaveAddr {
    size_t io_base0, ring_base0;
    unsigned io_off, ring_off;
};
__device__ __forceinline__ WaveAddr wave_addr(const ChainArgs &a, size_t c) {
    const unsigned c0 = __builtin_amdgcn_readfirstlane((unsigned)c);
    const size_t tile0 = c0 >> a.w_shift, cw0 = c0 & a.w_mask;
    const size_t tile = c >> a.w_shift, cw = c & a.w_mask;
    WaveAddr w;
    w.io_base0 = tile0 * a.io_tile_stride + cw0;
    w.ring_base0 = tile0 * RING_GROUP_ROWS * (size_t)a.ld + cw0;
    w.io_off = (unsigned)((tile * a.io_tile_stride + cw) - w.io_base0) * 4u;
    w.ring_off = (unsigned)((tile * RING_GROUP_ROWS * (size_t)a.ld + cw) - w.ring_base0) * 4u;
    return w;
}
// uniform base + 32-bit BYTE offset of the lane: the shape the global_load/store "saddr + voffset" form needs
__device__ __forceinline__ const float *lane_ptr(const float *base, unsigned byte_off) {
    return (const float *)((const char *)base + byte_off);
}
__device__ __forceinline__ float *lane_ptr(float *base, unsigned byte_off) { return (float *)((char *)base + byte_off); }


// Pin a wave-uniform pointer into scalar registers (the compiler otherwise may keep it in a VGPR pair per lane).
__device__ __forceinline__ float *uniform_ptr(float *p) {
    const unsigned long long v = (unsigned long long)p;
    const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)v), hi = __builtin_amdgcn_readfirstlane((unsigned)(v >> 32));
    return (float *)(((unsigned long long)hi << 32) | lo);
}
// The group pointers a span of at most 128 consecutive ring rows starting at row pos + f0 can need: the group of its first
// row, the one after it, and group 0 (rows after the wrap at D).  Three scalar loads issued together -- one table read PER
// ROW, each waited for before the row's address could be formed, cost a wave 32 x ~90 ns before its first tap load was
// even issued (tools/ts_timeline.py: 2.9 us at the head of the time-sliced kernel).
// (the time-sliced kernel, whose blocks are exactly 128 frames, takes the three pointers from its arguments instead: ring_groups_host)
__device__ __forceinline__ RingGroups ring_groups_host(const SlotArgs &s) { return RingGroups{s.g_a, s.g_b, s.g_0, s.g_ia}; }
__device__ __forceinline__ RingGroups ring_groups(const SlotArgs &s, const Ctx &cx) {
    unsigned r = s.pos + cx.f0;               // < 2*D: host keeps pos < D, nframes <= D
    r = r >= s.D ? r - s.D : r;
    // the group table is never written by a kernel: read it through the constant address space so the loads
    // are scalar ones whatever stores precede them
    typedef float *fptr_t;
    const __attribute__((address_space(4))) fptr_t *tab = (const __attribute__((address_space(4))) fptr_t *)s.groups;
    const unsigned gia = __builtin_amdgcn_readfirstlane(r >> 7), glast = __builtin_amdgcn_readfirstlane((s.D - 1) >> 7);
    RingGroups g;
    g.gia = gia;
    g.ga = tab[gia];
    g.gb = tab[gia < glast ? gia + 1 : glast];
    g.g0 = tab[0];
    return g;
}
// wave-uniform base of ring row (pos + f0 + f), f < 128, for this wave's first channel
__device__ __forceinline__ float *ring_row(const SlotArgs &s, const Ctx &cx, const RingGroups &g, int f) {
    unsigned r = s.pos + cx.f0 + f;
    r = r >= s.D ? r - s.D : r;
    const unsigned gi = __builtin_amdgcn_readfirstlane(r >> 7);
    float *gb = gi == g.gia ? g.ga : (gi == g.gia + 1 ? g.gb : g.g0);
    return uniform_ptr(gb + cx.ring_base0 + (size_t)(r & 127u) * cx.ld);
}
// ---- rows through a buffer descriptor -------------------------------------------------------------------------------------
// A wave of the time-sliced kernel touches 4 x 32 rows (samples in, taps, ring rows out, samples out) whose addresses are a
// wave-uniform row base + one per-lane byte offset.  As 64-bit pointers the compiler formed every row's address with a vector
// 64-bit add, kept the 64 row bases alive in SGPR pairs across the whole chain for the stores, and spilled them into VGPR
// lanes: ~570 of the 3-node kernel's 3800 vector-ALU slots and ~470 of the 5-node kernel's 2750 were v_lshl_add_u64 /
// v_writelane / v_readlane (round 4, ISA count).  A buffer instruction takes the base as a descriptor in SGPRs, the lane's
// offset in ONE VGPR for all rows and the row's byte offset as a scalar operand: no vector ALU work per row at all.
// Lanes of a guarded launch that are out of range get an offset beyond the descriptor's extent: their loads return 0 and
// their stores are dropped, branch-free.  Row offsets are 32-bit and relative to the chunk's (slice's) first row: at most
// 32 rows x 4 N bytes in the frame-major layout, i.e. fine up to 16 M channels per engine; 128 KiB in the tiled one.
constexpr unsigned ROW_OOB = 0x80000000u;
constexpr int BUF_AUX_NT = 2;
__device__ __forceinline__ __amdgpu_buffer_rsrc_t row_rsrc(const float *base) {
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(base), 0, 0x7fffffff, 0x00020000);
}
template <int CPL, int STREAM>
__device__ __forceinline__ void load_row(__amdgpu_buffer_rsrc_t r, unsigned voff, unsigned soff, float (&v)[CPL]) {
    static_assert(CPL == 1 || CPL == 2 || CPL == 4, "one, two or four channels per lane");
    constexpr int aux = nt_for(STREAM) ? BUF_AUX_NT : 0;
    if constexpr (CPL == 1) {
        v[0] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(r, (int)voff, (int)soff, aux));
    } else if constexpr (CPL == 2) {
        const dspfx_u32x2 t = __builtin_amdgcn_raw_buffer_load_b64(r, (int)voff, (int)soff, aux);
        v[0] = __uint_as_float(t.x);
        v[1] = __uint_as_float(t.y);
    } else {
        const dspfx_u32x4 t = __builtin_amdgcn_raw_buffer_load_b128(r, (int)voff, (int)soff, aux);
        v[0] = __uint_as_float(t.x);
        v[1] = __uint_as_float(t.y);
        v[2] = __uint_as_float(t.z);
        v[3] = __uint_as_float(t.w);
    }
}
template <int CPL, int STREAM>
__device__ __forceinline__ void store_row(__amdgpu_buffer_rsrc_t r, unsigned voff, unsigned soff, const float (&v)[CPL]) {
    constexpr int aux = nt_for(STREAM) ? BUF_AUX_NT : 0;
    if constexpr (CPL == 1) {
        __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(v[0]), r, (int)voff, (int)soff, aux);
    } else if constexpr (CPL == 2) {
        dspfx_u32x2 t;
        t.x = __float_as_uint(v[0]);
        t.y = __float_as_uint(v[1]);
        __builtin_amdgcn_raw_buffer_store_b64(t, r, (int)voff, (int)soff, aux);
    } else {
        dspfx_u32x4 t;
        t.x = __float_as_uint(v[0]);
        t.y = __float_as_uint(v[1]);
        t.z = __float_as_uint(v[2]);
        t.w = __float_as_uint(v[3]);
        __builtin_amdgcn_raw_buffer_store_b128(t, r, (int)voff, (int)soff, aux);
    }
}
// ring row (pos + f0 + f) as (descriptor of its group at this wave's first channel, byte offset of the row inside the group)
struct RowRef {
    __amdgpu_buffer_rsrc_t rsrc;
    unsigned soff;
};
// pos0: s.pos + cx.f0, handed in by the caller -- once as it is for the loads and once through opaque_u32 for the stores, so that
// the compiler forms the 32 descriptors AGAIN at the stores instead of keeping them alive across the whole chain
__device__ __forceinline__ unsigned opaque_u32(unsigned x) {
    asm volatile("" : "+s"(x));
    return x;
}
__device__ __forceinline__ RowRef ring_row_ref(const SlotArgs &s, const Ctx &cx, const RingGroups &g, unsigned pos0, int f) {
    unsigned r = pos0 + f;
    r = r >= s.D ? r - s.D : r;
    const unsigned gi = r >> 7;
    float *gb = gi == g.gia ? g.ga : (gi == g.gia + 1 ? g.gb : g.g0);
    return RowRef{row_rsrc(gb + cx.ring_base0), (r & 127u) * (unsigned)cx.ld * 4u};
}
template <int F, int CPL, bool GUARD>
__device__ __forceinline__ void ring_prefetch(const SlotArgs &s, const Ctx &cx, RingPre<F, CPL> &pre) {
    static_assert(F <= 128, "ring_groups covers spans of at most one group length");
    const RingGroups g = ring_groups(s, cx);
    const unsigned voff = (GUARD && !cx.active) ? ROW_OOB : cx.ring_off;
#pragma unroll
    for (int f = 0; f < F; ++f) {
        pre.row[f] = ring_row(s, cx, g, f);
        load_row<CPL, S_RING_LD>(row_rsrc(pre.row[f]), voff, 0u, pre.tap[f]);
    }
}
// A tap of frame cx.f0 + f that was written before the ring's last clear reads as the +0.0 the reference's new ring holds
// (SlotArgs::zero_rows).  The compare is wave-uniform; the product with `decay` is still formed (0.0 * decay keeps the
// reference's signs and NaNs).  A branch of its own so that the steady state pays one scalar compare per chunk.
template <int F, int CPL>
__device__ __forceinline__ void ring_zero_cleared(const SlotArgs &s, const Ctx &cx, float (&tap)[F][CPL]) {
#ifdef DSPFX_NO_LAZY_CLEAR      // A/B builds only (what the steady state pays for the compare): the clear is NOT applied
    return;
#endif
    const unsigned zr = s.zero_rows;
    if (zr > cx.f0) {
#pragma unroll
        for (int f = 0; f < F; ++f) {
            const bool cleared = cx.f0 + (unsigned)f < zr;
#pragma unroll
            for (int j = 0; j < CPL; ++j) tap[f][j] = cleared ? 0.0f : tap[f][j];
        }
    }
}
template <int F, int CPL, bool GUARD>
__device__ __forceinline__ void ring_apply(const SlotArgs &s, float (&v)[F][CPL], RingPre<F, CPL> &pre, const Ctx &cx) {
    const float decay = s.p[0];
    ring_zero_cleared<F, CPL>(s, cx, pre.tap);
    const unsigned voff = (GUARD && !cx.active) ? ROW_OOB : cx.ring_off;
#pragma unroll
    for (int f = 0; f < F; ++f) {
#pragma unroll
        for (int j = 0; j < CPL; ++j) v[f][j] = v[f][j] + pre.tap[f][j] * decay;
        store_row<CPL, S_RING_ST>(row_rsrc(pre.row[f]), voff, 0u, v[f]);
    }
}

// Apply one node to a chunk v[F][CPL]; st[][] is the node's per-channel state.
template <int KIND, int MODE, int F, int CPL, bool GUARD, bool FAST>
__device__ __forceinline__ void apply_node(const SlotArgs &s, float (&v)[F][CPL], float (&st)[4][CPL],
                                           const Ctx &cx) {
    if constexpr (KIND == K_GAIN) {               // gain.rs:33-37
        const float level = s.p[0];
#pragma unroll
        for (int f = 0; f < F; ++f)
#pragma unroll
            for (int j = 0; j < CPL; ++j) v[f][j] = v[f][j] * level;
    } else if constexpr (KIND == K_BIQUAD) {      // biquad.rs:87 -> DirectForm1::run
        const float a1 = s.p[0], a2 = s.p[1], b0 = s.p[2], b1 = s.p[3], b2 = s.p[4];
#pragma unroll
        for (int f = 0; f < F; ++f)
#pragma unroll
            for (int j = 0; j < CPL; ++j) {
                const float x = v[f][j];
                const float y = b0 * x + b1 * st[0][j] + b2 * st[1][j] - a1 * st[2][j] - a2 * st[3][j];
                st[1][j] = st[0][j];
                st[0][j] = x;
                st[3][j] = st[2][j];
                st[2][j] = y;
                v[f][j] = y;
            }
    } else if constexpr (KIND == K_LOW_PASS) {    // low_pass.rs:36-39
        const float r = s.p[0];
#pragma unroll
        for (int f = 0; f < F; ++f)
#pragma unroll
            for (int j = 0; j < CPL; ++j) {
                const float y = v[f][j] * (1.0f - r) + r * st[0][j];
                st[0][j] = y;
                v[f][j] = y;
            }
    } else if constexpr (KIND == K_HIGH_PASS) {   // high_pass.rs:36-39
        const float r = s.p[0];
#pragma unroll
        for (int f = 0; f < F; ++f)
#pragma unroll
            for (int j = 0; j < CPL; ++j) {
                const float z = v[f][j] * (1.0f - r) + r * st[0][j];
                st[0][j] = z;
                v[f][j] = v[f][j] - z;
            }
    } else if constexpr (KIND == K_REVERB) {      // reverb.rs:86-103: y = x + tap*decay; ring <- y
        RingPre<F, CPL> pre;
        ring_prefetch<F, CPL, GUARD>(s, cx, pre);
        ring_apply<F, CPL, GUARD>(s, v, pre, cx);
    } else if constexpr (KIND == K_DISTORT) {     // distort.rs:176-194 (Fuzz has its own kernel)
        const float level = s.p[0];
        if (level < 0.001f) return;               // every mode: `if level < 0.001 { return sample }`
#pragma unroll
        for (int f = 0; f < F; ++f)
#pragma unroll
            for (int j = 0; j < CPL; ++j) v[f][j] = distort1<MODE, FAST>(v[f][j], level, s.rc, cx.third_rc);
    } else if constexpr (KIND == K_OVERDRIVE) {   // overdrive.rs:58-72
        if (s.p[2] < 0.001f) return;              // overdrive.rs:32-34
#pragma unroll
        for (int f = 0; f < F; ++f)
#pragma unroll
            for (int j = 0; j < CPL; ++j) v[f][j] = overdrive1(v[f][j], s.p[0], s.p[1], s.p[2]);
    } else if constexpr (KIND == K_CHEBYSHEV) {   // chebyshev.rs:52-62
        const float tp = tanh_cr(s.p[0]), tn = tanh_cr(s.p[1]);
#pragma unroll
        for (int f = 0; f < F; ++f)
#pragma unroll
            for (int j = 0; j < CPL; ++j) v[f][j] = chebyshev1(v[f][j], s.p[0], s.p[1], tp, tn);
    } else if constexpr (KIND == K_ENVELOPE) {    // envelope.rs:34-52, dasp_envelope Detector::next + dasp_peak full_wave
        const float ga = s.p[0], gr = s.p[1];     // attack / release gains, computed on the host (powf)
#pragma unroll
        for (int f = 0; f < F; ++f) {
#pragma unroll
            for (int j = 0; j < CPL; ++j) {
                const float x = v[f][j];
                const float d = x < 0.0f ? -x : x;
                const float l = st[0][j];
                const float gain = l < d ? ga : gr;
                const float diff = l + (-d);
                st[0][j] = d + diff * gain;
                v[f][j] = st[0][j];
            }
        }
    } else if constexpr (KIND == K_SIGNAL_GEN) {  // signal_gen.rs:111-128: a source, the input is ignored
        // st[0] = clock (persisted), st[1] = block-local total; the clock wraps at every 128-frame block end
#pragma unroll
        for (int f = 0; f < F; ++f) {
#pragma unroll
            for (int j = 0; j < CPL; ++j) {
                if constexpr (MODE == G_SQUARE_OR_CONST) {
                    const bool is_const = s.mode == G_CONSTANT;
                    const float sq = signal1<G_SQUARE>(st[0][j], st[1][j], is_const ? 0.0f : s.p[1], s.p[0]);
                    v[f][j] = is_const ? s.p[0] : sq;
                } else {
                    v[f][j] = signal1<MODE>(st[0][j], st[1][j], s.p[1], s.p[0]);
                }
            }
            // Constant leaves the clock alone (signal_gen.rs:106-108): under G_SQUARE_OR_CONST its total stays 0 and
            // fmod(clock + 0, 1) == clock for the clock's range [0, 1)
            if (MODE != G_CONSTANT && ((cx.f0 + f + 1) & 127u) == 0) {
#pragma unroll
                for (int j = 0; j < CPL; ++j) {
                    st[0][j] = fmodf(st[0][j] + st[1][j], 1.0f);   // signal_gen.rs:66-67
                    st[1][j] = 0.0f;
                }
            }
        }
    } else if constexpr (KIND == K_ADD || KIND == K_MIX) {   // add.rs:29-33, mix.rs:41-46
        const float ratio = s.p[0];
#pragma unroll
        for (int f = 0; f < F; ++f) {
            float b[CPL];
            if (cx.side) {
                load_vec<CPL, GUARD, S_IN>(lane_ptr(cx.side + cx.io_base0 + (size_t)(cx.f0 + f) * cx.ld, cx.io_off), b, cx.active);
                if (cx.side_hop & 1) {
#pragma unroll
                    for (int j = 0; j < CPL; ++j) b[j] = link_hop<FAST>(b[j], cx.hop_div, cx.hop_rc);
                }
            } else {
#pragma unroll
                for (int j = 0; j < CPL; ++j) b[j] = 0.0f;    // unconnected port: zeros (node.rs:288)
            }
#pragma unroll
            for (int j = 0; j < CPL; ++j) {
                if constexpr (KIND == K_ADD) v[f][j] = v[f][j] + b[j];
                else v[f][j] = (b[j] * ratio) + (v[f][j] * (1.0f - ratio));
            }
        }
    }
}

// ---- control ports ----------------------------------------------------------------------
// Per-sample value of slider k (dsp-stuff-derive/src/lib.rs:135-153): a connected port maps the
// signal [-1,1] -> [lo,hi] per sample and latches the first value of every 128-frame block into
// the slider (per channel here); an unconnected port fills with the slider value.
// signal [-1, 1] -> slider range [lo, hi] (dsp-stuff-derive/src/lib.rs:139-146)
__device__ __forceinline__ float slider_map(float x, float lo, float hi) {
    const float y = (x + 1.0f) / 2.0f;
    float z = y < 0.0f ? 0.0f : y;          // f32::clamp(0.0, 1.0): NaN stays NaN
    z = z > 1.0f ? 1.0f : z;
    return lo + (hi - lo) * z;
}
template <int F, int CPL, bool GUARD, bool FAST>
__device__ __forceinline__ void slider_values(const SlotArgs &s, int k, float lo, float hi, const Ctx &cx,
                                              float (&p)[F][CPL]) {
    if (s.ctl[k]) {
#pragma unroll
        for (int f = 0; f < F; ++f) {
            load_vec<CPL, GUARD, S_IN>(lane_ptr(s.ctl[k] + cx.io_base0 + (size_t)(cx.f0 + f) * cx.ld, cx.io_off), p[f], cx.active);
#pragma unroll
            for (int j = 0; j < CPL; ++j) {
                float x = p[f][j];
                if (cx.side_hop & 2) x = link_hop<FAST>(x, cx.hop_div, cx.hop_rc);   // the control link's collect_and_average
                p[f][j] = slider_map(x, lo, hi);
            }
        }
        if ((cx.f0 & 127u) == 0) store_vec<CPL, GUARD>(s.latch[k] + cx.c, p[0], cx.active);   // lib.rs:148
    } else if (s.latch_valid & (1 << k)) {
        float l[CPL];
        load_vec<CPL, GUARD>(s.latch[k] + cx.c, l, cx.active);
#pragma unroll
        for (int f = 0; f < F; ++f)
#pragma unroll
            for (int j = 0; j < CPL; ++j) p[f][j] = l[j];
    } else {
#pragma unroll
        for (int f = 0; f < F; ++f)
#pragma unroll
            for (int j = 0; j < CPL; ++j) p[f][j] = s.p[k];
    }
}

// Nodes with `as_input` sliders, evaluated with per-sample slider values (IEEE division: the
// divisor is no longer a wave-uniform constant).  The `*_mod_core` functions take the slider values as arrays:
// apply_node_mod fills them from control buffers in memory, the whole-graph kernel (graph_kernel.hip.h) from registers.
template <int F, int CPL>
__device__ __forceinline__ void gain_mod_core(float (&v)[F][CPL], const float (&lv)[F][CPL]) {   // gain.rs:27-37
#pragma unroll
    for (int f = 0; f < F; ++f)
#pragma unroll
        for (int j = 0; j < CPL; ++j) v[f][j] = v[f][j] * lv[f][j];
}
template <int MODE, int F, int CPL>
__device__ __forceinline__ void distort_mod_core(float (&v)[F][CPL], const float (&lv)[F][CPL]) {   // distort.rs:176-194
#pragma unroll
    for (int f = 0; f < F; ++f)
#pragma unroll
        for (int j = 0; j < CPL; ++j)
            if (!(lv[f][j] < 0.001f)) v[f][j] = distort1<MODE, false>(v[f][j], lv[f][j], 0.0, 0.0);
}
template <int F, int CPL>
__device__ __forceinline__ void overdrive_mod_core(float (&v)[F][CPL], const float (&bo)[F][CPL], const float (&dr)[F][CPL],
                                                   const float (&lv)[F][CPL]) {   // overdrive.rs:58-72
#pragma unroll
    for (int f = 0; f < F; ++f)
#pragma unroll
        for (int j = 0; j < CPL; ++j)
            if (!(lv[f][j] < 0.001f)) v[f][j] = overdrive1(v[f][j], bo[f][j], dr[f][j], lv[f][j]);
}
template <int MODE, int F, int CPL>
__device__ __forceinline__ void siggen_mod_core(const SlotArgs &s, float (&v)[F][CPL], float (&st)[4][CPL], const float (&am)[F][CPL],
                                                const float (&fr)[F][CPL], const Ctx &cx) {   // signal_gen.rs:111-128
#pragma unroll
    for (int f = 0; f < F; ++f) {
#pragma unroll
        for (int j = 0; j < CPL; ++j) {
            if constexpr (MODE == G_SQUARE_OR_CONST) {
                const bool is_const = s.mode == G_CONSTANT;
                const float sq = signal1<G_SQUARE>(st[0][j], st[1][j], is_const ? 0.0f : fr[f][j], am[f][j]);
                v[f][j] = is_const ? am[f][j] : sq;
            } else {
                v[f][j] = signal1<MODE>(st[0][j], st[1][j], fr[f][j], am[f][j]);
            }
        }
        if (MODE != G_CONSTANT && ((cx.f0 + f + 1) & 127u) == 0) {
#pragma unroll
            for (int j = 0; j < CPL; ++j) {
                st[0][j] = fmodf(st[0][j] + st[1][j], 1.0f);
                st[1][j] = 0.0f;
            }
        }
    }
}

template <int KIND, int MODE, int F, int CPL, bool GUARD, bool FAST>
__device__ __forceinline__ void apply_node_mod(const SlotArgs &s, float (&v)[F][CPL], float (&st)[4][CPL], const Ctx &cx) {
    if constexpr (KIND == K_GAIN) {                 // slider 0..=10
        float lv[F][CPL];
        slider_values<F, CPL, GUARD, FAST>(s, 0, 0.0f, 10.0f, cx, lv);
        gain_mod_core<F, CPL>(v, lv);
    } else if constexpr (KIND == K_DISTORT) {       // slider 0..=30
        float lv[F][CPL];
        slider_values<F, CPL, GUARD, FAST>(s, 0, 0.0f, 30.0f, cx, lv);
        distort_mod_core<MODE, F, CPL>(v, lv);
    } else if constexpr (KIND == K_OVERDRIVE) {     // sliders boost 0..=30, drive 0..=1, level 0..=1
        float bo[F][CPL], dr[F][CPL], lv[F][CPL];
        slider_values<F, CPL, GUARD, FAST>(s, 0, 0.0f, 30.0f, cx, bo);
        slider_values<F, CPL, GUARD, FAST>(s, 1, 0.0f, 1.0f, cx, dr);
        slider_values<F, CPL, GUARD, FAST>(s, 2, 0.0f, 1.0f, cx, lv);
        overdrive_mod_core<F, CPL>(v, bo, dr, lv);
    } else if constexpr (KIND == K_SIGNAL_GEN) {    // sliders amplitude -1..=1, frequency 0.1..=20000
        float am[F][CPL], fr[F][CPL];
        slider_values<F, CPL, GUARD, FAST>(s, 0, -1.0f, 1.0f, cx, am);
        slider_values<F, CPL, GUARD, FAST>(s, 1, 0.1f, 20000.0f, cx, fr);
        siggen_mod_core<MODE, F, CPL>(s, v, st, am, fr, cx);
    } else if constexpr (KIND == K_MIX) {           // mix.rs:33-46, slider 0..=1
        float ra[F][CPL];
        slider_values<F, CPL, GUARD, FAST>(s, 0, 0.0f, 1.0f, cx, ra);
#pragma unroll
        for (int f = 0; f < F; ++f) {
            float b[CPL];
            if (cx.side) {
                load_vec<CPL, GUARD, S_IN>(lane_ptr(cx.side + cx.io_base0 + (size_t)(cx.f0 + f) * cx.ld, cx.io_off), b, cx.active);
                if (cx.side_hop & 1) {
#pragma unroll
                    for (int j = 0; j < CPL; ++j) b[j] = link_hop<FAST>(b[j], cx.hop_div, cx.hop_rc);
                }
            } else {
#pragma unroll
                for (int j = 0; j < CPL; ++j) b[j] = 0.0f;
            }
#pragma unroll
            for (int j = 0; j < CPL; ++j) v[f][j] = (b[j] * ra[f][j]) + (v[f][j] * (1.0f - ra[f][j]));
        }
    }
}

template <int F, int CPL, bool GUARD, bool FAST>
__device__ __forceinline__ void apply_distort_mod_dyn(const SlotArgs &s, float (&v)[F][CPL], float (&st)[4][CPL], const Ctx &cx) {
    switch (s.mode) {
    case D_HARD_CLIP: apply_node_mod<K_DISTORT, D_HARD_CLIP, F, CPL, GUARD, FAST>(s, v, st, cx); break;
    case D_SOFT_CLIP: apply_node_mod<K_DISTORT, D_SOFT_CLIP, F, CPL, GUARD, FAST>(s, v, st, cx); break;
    case D_TANH: apply_node_mod<K_DISTORT, D_TANH, F, CPL, GUARD, FAST>(s, v, st, cx); break;
    case D_RECIP_SOFT_CLIP: apply_node_mod<K_DISTORT, D_RECIP_SOFT_CLIP, F, CPL, GUARD, FAST>(s, v, st, cx); break;
    case D_SIN: apply_node_mod<K_DISTORT, D_SIN, F, CPL, GUARD, FAST>(s, v, st, cx); break;
    case D_ATAN: apply_node_mod<K_DISTORT, D_ATAN, F, CPL, GUARD, FAST>(s, v, st, cx); break;
    case D_SQUARE: apply_node_mod<K_DISTORT, D_SQUARE, F, CPL, GUARD, FAST>(s, v, st, cx); break;
    case D_CHEBYSHEV4: apply_node_mod<K_DISTORT, D_CHEBYSHEV4, F, CPL, GUARD, FAST>(s, v, st, cx); break;
    default: break;
    }
}

template <int F, int CPL, bool FAST>
__device__ __forceinline__ void apply_hop(float (&v)[F][CPL], float div, double rc) {
#pragma unroll
    for (int f = 0; f < F; ++f)
#pragma unroll
        for (int j = 0; j < CPL; ++j) v[f][j] = link_hop<FAST>(v[f][j], div, rc);
}

template <int F, int CPL, bool GUARD, bool FAST, bool LIBM>
__device__ __forceinline__ void apply_distort_dyn(const SlotArgs &s, float (&v)[F][CPL], float (&st)[4][CPL],
                                                  const Ctx &cx) {
    switch (s.mode) {
    case D_HARD_CLIP: apply_node<K_DISTORT, D_HARD_CLIP, F, CPL, GUARD, FAST>(s, v, st, cx); break;
    case D_SOFT_CLIP: apply_node<K_DISTORT, D_SOFT_CLIP, F, CPL, GUARD, FAST>(s, v, st, cx); break;
    case D_TANH: if constexpr (LIBM) apply_node<K_DISTORT, D_TANH, F, CPL, GUARD, FAST>(s, v, st, cx); break;
    case D_RECIP_SOFT_CLIP: apply_node<K_DISTORT, D_RECIP_SOFT_CLIP, F, CPL, GUARD, FAST>(s, v, st, cx); break;
    case D_SIN: if constexpr (LIBM) apply_node<K_DISTORT, D_SIN, F, CPL, GUARD, FAST>(s, v, st, cx); break;
    case D_ATAN: if constexpr (LIBM) apply_node<K_DISTORT, D_ATAN, F, CPL, GUARD, FAST>(s, v, st, cx); break;
    case D_SQUARE: apply_node<K_DISTORT, D_SQUARE, F, CPL, GUARD, FAST>(s, v, st, cx); break;
    case D_CHEBYSHEV4: apply_node<K_DISTORT, D_CHEBYSHEV4, F, CPL, GUARD, FAST>(s, v, st, cx); break;
    default: break;
    }
}

// One slot: static signature => everything folds at compile time; SIG_DYN => a
// wave-uniform switch (scalar branches, no divergence).
template <int SIG, int F, int CPL, bool GUARD, bool FAST, bool MOD = false, bool LIBM = true>
__device__ __forceinline__ void run_slot(const SlotArgs &s, float (&v)[F][CPL], float (&st)[4][CPL],
                                         const Ctx &cx) {
    if constexpr (SIG == SIG_NONE) {
        return;
    } else if constexpr (SIG == SIG_DYN) {
        if (s.hop) apply_hop<F, CPL, FAST>(v, cx.hop_div, cx.hop_rc);
        if (MOD && (s.ctl[0] || s.ctl[1] || s.ctl[2] || s.latch_valid)) {   // modulated / latched sliders
            switch (s.kind) {
            case K_GAIN: apply_node_mod<K_GAIN, 0, F, CPL, GUARD, FAST>(s, v, st, cx); return;
            case K_DISTORT: apply_distort_mod_dyn<F, CPL, GUARD, FAST>(s, v, st, cx); return;
            case K_OVERDRIVE: apply_node_mod<K_OVERDRIVE, 0, F, CPL, GUARD, FAST>(s, v, st, cx); return;
            case K_MIX: apply_node_mod<K_MIX, 0, F, CPL, GUARD, FAST>(s, v, st, cx); return;
            case K_SIGNAL_GEN:
                if (s.mode == G_SINE) apply_node_mod<K_SIGNAL_GEN, G_SINE, F, CPL, GUARD, FAST>(s, v, st, cx);
                else if (s.mode == G_TRIANGLE) apply_node_mod<K_SIGNAL_GEN, G_TRIANGLE, F, CPL, GUARD, FAST>(s, v, st, cx);
                else apply_node_mod<K_SIGNAL_GEN, G_SQUARE_OR_CONST, F, CPL, GUARD, FAST>(s, v, st, cx);
                return;
            default: break;
            }
        }
        switch (s.kind) {
        case K_GAIN: apply_node<K_GAIN, 0, F, CPL, GUARD, FAST>(s, v, st, cx); break;
        case K_BIQUAD: apply_node<K_BIQUAD, 0, F, CPL, GUARD, FAST>(s, v, st, cx); break;
        case K_LOW_PASS: apply_node<K_LOW_PASS, 0, F, CPL, GUARD, FAST>(s, v, st, cx); break;
        case K_HIGH_PASS: apply_node<K_HIGH_PASS, 0, F, CPL, GUARD, FAST>(s, v, st, cx); break;
        case K_REVERB: apply_node<K_REVERB, 0, F, CPL, GUARD, FAST>(s, v, st, cx); break;
        case K_DISTORT: apply_distort_dyn<F, CPL, GUARD, FAST, LIBM>(s, v, st, cx); break;
        case K_OVERDRIVE: if constexpr (LIBM) apply_node<K_OVERDRIVE, 0, F, CPL, GUARD, FAST>(s, v, st, cx); break;
        case K_CHEBYSHEV: if constexpr (LIBM) apply_node<K_CHEBYSHEV, 0, F, CPL, GUARD, FAST>(s, v, st, cx); break;
        case K_ADD: apply_node<K_ADD, 0, F, CPL, GUARD, FAST>(s, v, st, cx); break;
        case K_MIX: apply_node<K_MIX, 0, F, CPL, GUARD, FAST>(s, v, st, cx); break;
        case K_ENVELOPE: apply_node<K_ENVELOPE, 0, F, CPL, GUARD, FAST>(s, v, st, cx); break;
        case K_SIGNAL_GEN:
            if constexpr (LIBM) {
                if (s.mode == G_SINE) apply_node<K_SIGNAL_GEN, G_SINE, F, CPL, GUARD, FAST>(s, v, st, cx);
                else if (s.mode == G_TRIANGLE) apply_node<K_SIGNAL_GEN, G_TRIANGLE, F, CPL, GUARD, FAST>(s, v, st, cx);
                else apply_node<K_SIGNAL_GEN, G_SQUARE_OR_CONST, F, CPL, GUARD, FAST>(s, v, st, cx);
            }
            break;
        default: break;
        }
    } else {
        if constexpr (sig_hop(SIG)) apply_hop<F, CPL, FAST>(v, cx.hop_div, cx.hop_rc);
        constexpr int K = sig_kind(SIG);
        constexpr bool has_ports = K == K_GAIN || K == K_DISTORT || K == K_OVERDRIVE || K == K_MIX || K == K_SIGNAL_GEN;
        if constexpr (MOD && has_ports) {          // specialised kernel with control ports (wave-uniform test)
            if (s.ctl[0] || s.ctl[1] || s.ctl[2] || s.latch_valid) {
                apply_node_mod<K, sig_mode(SIG), F, CPL, GUARD, FAST>(s, v, st, cx);
                return;
            }
        }
        apply_node<K, sig_mode(SIG), F, CPL, GUARD, FAST>(s, v, st, cx);
    }
}

// rows of per-channel state a kind keeps in registers / LDS during a launch, and how many of them
// persist in HBM between launches (SIGNAL_GEN: the clock persists, the block-local total does not)
__host__ __device__ __forceinline__ constexpr int kind_nstate(int k) {
    return k == K_BIQUAD ? 4 : (k == K_LOW_PASS || k == K_HIGH_PASS || k == K_ENVELOPE) ? 1 : k == K_SIGNAL_GEN ? 2 : 0;
}
__host__ __device__ __forceinline__ constexpr int kind_npersist(int k) { return k == K_SIGNAL_GEN ? 1 : kind_nstate(k); }
template <int SIG>
__device__ __forceinline__ int slot_nstate(const SlotArgs &s) {
    if constexpr (SIG == SIG_NONE) return 0;
    else if constexpr (SIG == SIG_DYN) return kind_nstate(s.kind);
    else return kind_nstate(sig_kind(SIG));
}
template <int SIG>
__device__ __forceinline__ int slot_npersist(const SlotArgs &s) {
    if constexpr (SIG == SIG_NONE) return 0;
    else if constexpr (SIG == SIG_DYN) return kind_npersist(s.kind);
    else return kind_npersist(sig_kind(SIG));
}
// A launch that ends inside a 128-frame block closes the generator's block (the caller's blocks are the
// reference's blocks): clock = (clock + total) % 1.0
template <int CPL>
__device__ __forceinline__ void signal_gen_close_block(const SlotArgs &s, float (&st)[4][CPL], unsigned nframes) {
    if (s.kind == K_SIGNAL_GEN && s.mode != G_CONSTANT && (nframes & 127u)) {
#pragma unroll
        for (int j = 0; j < CPL; ++j) {
            st[0][j] = fmodf(st[0][j] + st[1][j], 1.0f);
            st[1][j] = 0.0f;
        }
    }
}

template <int SIG, int CPL, bool GUARD>
__device__ __forceinline__ void load_state(const SlotArgs &s, float (&st)[4][CPL], size_t c, size_t N, bool active) {
    const int n = slot_npersist<SIG>(s);
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        if (k < n) load_vec<CPL, GUARD>(s.state + (size_t)k * N + c, st[k], active);
        else {
#pragma unroll
            for (int j = 0; j < CPL; ++j) st[k][j] = 0.0f;
        }
    }
}
template <int SIG, int CPL, bool GUARD>
__device__ __forceinline__ void store_state(const SlotArgs &s, const float (&st)[4][CPL], size_t c, size_t N, bool active) {
    const int n = slot_npersist<SIG>(s);
#pragma unroll
    for (int k = 0; k < 4; ++k)
        if (k < n) store_vec<CPL, GUARD>(s.state + (size_t)k * N + c, st[k], active);
}

// Wave-level reduce-scatter of F per-lane values over the 64 lanes: after it,
// r[0] of lane l holds the wave total of frame  mixbus_frame_of_lane<F>(l).
// 2F-2+max(0,6-log2F) adds instead of 6F; fixed order => deterministic.
// Compile-time recursion keeps every r[] index a constant (registers, no select chains).
// All lane exchanges are VALU operations (no LDS crossbar round trips): gfx950's
// v_permlane32_swap / v_permlane16_swap exchange the halves of two registers in ONE instruction, which
// is exactly a reduce-scatter step (lower lanes keep `lo` and receive the partner's `lo`, upper lanes
// keep `hi` and receive the partner's `hi`); the steps inside a row of 16 use DPP operands
// (row_ror:8, row_half_mirror, quad_perm) folded into the v_add_f32.
typedef unsigned int dspfx_u2 __attribute__((ext_vector_type(2)));
template <int O>
__device__ __forceinline__ float lane_partner(float x) {
    // value of the lane paired with this one across lane-index bit O (any pairing across that bit works)
    constexpr int ctrl = O == 8 ? 0x128      /* row_ror:8          l <-> l^8      */
                       : O == 4 ? 0x141      /* row_half_mirror    l <-> l^7      */
                       : O == 2 ? 0x4e       /* quad_perm:[2,3,0,1] l <-> l^2     */
                                : 0xb1;      /* quad_perm:[1,0,3,2] l <-> l^1     */
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(x), ctrl, 0xf, 0xf, false));
}
template <int O>
__device__ __forceinline__ float swap_add(float lo, float hi) {
    // lanes with bit O clear: lo(self) + lo(partner); lanes with bit O set: hi(partner) + hi(self)
    dspfx_u2 t;
    if constexpr (O == 32) t = __builtin_amdgcn_permlane32_swap(__float_as_uint(lo), __float_as_uint(hi), false, false);
    else t = __builtin_amdgcn_permlane16_swap(__float_as_uint(lo), __float_as_uint(hi), false, false);
    return __uint_as_float(t.x) + __uint_as_float(t.y);
}
template <int F, int NLIVE, int O>
__device__ __forceinline__ void rs_stage(float (&r)[F], int lane) {
    if constexpr (O >= 1) {
        if constexpr (NLIVE > 1) {
            constexpr int H = NLIVE / 2;
            if constexpr (O >= 16) {
#pragma unroll
                for (int i = 0; i < H; ++i) r[i] = swap_add<O>(r[i], r[H + i]);
            } else {
                const bool up = (lane & O) != 0;
#pragma unroll
                for (int i = 0; i < H; ++i) {
                    const float lo = r[i], hi = r[H + i];
                    const float keep = up ? hi : lo;
                    const float send = up ? lo : hi;
                    r[i] = keep + lane_partner<O>(send);
                }
            }
            rs_stage<F, H, O / 2>(r, lane);
        } else {
            if constexpr (O >= 16) r[0] = swap_add<O>(r[0], r[0]);
            else r[0] = r[0] + lane_partner<O>(r[0]);
            rs_stage<F, 1, O / 2>(r, lane);
        }
    }
}
template <int F>
__device__ __forceinline__ void wave_reduce_scatter(float (&r)[F], int lane) {
    static_assert((F & (F - 1)) == 0 && F <= 64, "F must be a power of two");
    rs_stage<F, F, 32>(r, lane);
}
template <int F>
__device__ __forceinline__ int mixbus_frame_of_lane(int lane) {
    int f = 0, n = F;
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) {
        if (n > 1) {
            n /= 2;
            if (lane & o) f += n;
        }
    }
    return f;
}
template <int F>
__device__ __forceinline__ bool mixbus_lane_writes(int lane) {
    // lanes that differ only in the bits consumed by the plain-butterfly tail hold duplicates
    int n = F, mask = 0;
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) {
        if (n > 1) n /= 2;
        else mask |= o;
    }
    return (lane & mask) == 0;
}

// Block -> work mapping.  The dispatcher places block b on XCD b % 8, so with the identity mapping the
// workgroups resident on one XCD work on channel tiles 8 apart: a regular 1 MiB stride that can alias
// onto few of that XCD's L2 channels (measured: TCP_TCR_TCP_STALL_CYCLES x6.4 and +18 % kernel time for
// unlucky physical placements, profiles/r01_placement.txt).  Remapped, XCD x owns blocks
// [x*nb/8, (x+1)*nb/8): neighbouring workgroups of an XCD touch neighbouring 128 KiB tiles.
// Placement only affects speed, never results (every block index is still covered exactly once).
__device__ __forceinline__ unsigned work_block(int remap) {
    const unsigned b = blockIdx.x, nb = gridDim.x;
    if (!remap || (nb & 7u)) return b;
    return (b & 7u) * (nb >> 3) + (b >> 3);
}

// ---- the fused chain kernel, statically specialised ---------------------------------
#define DSPFX_FOR_SLOTS(M) M(0) M(1) M(2) M(3) M(4) M(5) M(6) M(7)

template <int... SIGS> struct SigList {
    static constexpr int v[MAX_SLOTS] = {SIGS...};
};

// The bus' first stage.  Every wave reduces its chunk over its 64 lanes (reduce-scatter: one total per frame) and parks the
// totals in LDS; once a 128-frame segment of the block is complete in all waves of the workgroup, wave 0 adds the waves'
// rows in fixed order -- (w0 + w1) + (w2 + w3) -- and writes ONE row segment per workgroup (mixbus_flush).  A quarter
// of the rows for the later stages to read, and no global stores inside the chunk loop.  Two buffers, used in turn, so one
// barrier per segment suffices: nobody writes segment s + 2 before the barrier of segment s + 1, which wave 0 reaches
// only after it has read segment s.
constexpr unsigned MIX_SEG = 128;
struct MixStage {
    float row[2][WG / 64][MIX_SEG];
};
template <int F, int CPL>
__device__ __forceinline__ void mixbus_partial(MixStage &ms, const float (&v)[F][CPL], bool live, unsigned f0, int lane, int wave) {
    static_assert(MIX_SEG % F == 0, "a chunk never straddles two segments");
    float r[F];
#pragma unroll
    for (int f = 0; f < F; ++f) {
        float sacc = v[f][0];
#pragma unroll
        for (int j = 1; j < CPL; ++j) sacc = sacc + v[f][j];
        r[f] = live ? sacc : 0.0f;
    }
    wave_reduce_scatter<F>(r, lane);
    if (mixbus_lane_writes<F>(lane)) {
        const unsigned f = f0 + mixbus_frame_of_lane<F>(lane);
        ms.row[(f / MIX_SEG) & 1][wave][f % MIX_SEG] = r[0];
    }
}
// Frames [seg0, seg0 + len) are parked by every wave of the workgroup (nw of them are alive; they all call this).
// per_wave (ChainArgs::mix_per_wave): every live wave writes the row it parked itself, `row` being the workgroup's first.
template <class ARGS>
__device__ __forceinline__ void mixbus_flush(const ARGS &a, MixStage &ms, unsigned seg0, unsigned len, unsigned row, int nw, int wave, int lane, bool per_wave = false) {
    __syncthreads();
    const int buf = (seg0 / MIX_SEG) & 1;
    if (per_wave) {
        if (wave >= nw) return;
        float *dst = a.mixpart + (size_t)(row + (unsigned)wave) * a.nframes + seg0;
        for (unsigned f = lane; f < len; f += 64) {
            const float t = ms.row[buf][wave][f];
            if (a.mt_tickets) st_sc1(dst + f, t);
            else dst[f] = t;
        }
        return;
    }
    if (wave != 0) return;
    float *dst = a.mixpart + (size_t)row * a.nframes + seg0;
    for (unsigned f = lane; f < len; f += 64) {
        float t = ms.row[buf][0][f];
        if (nw > 1) t = t + ms.row[buf][1][f];
        if (nw > 2) {
            float u = ms.row[buf][2][f];
            if (nw > 3) u = u + ms.row[buf][3][f];
            t = t + u;
        }
        if (a.mt_tickets) st_sc1(dst + f, t);        // handed to another workgroup inside this launch: written through
        else dst[f] = t;
    }
}
// waves of this workgroup that cover channels (the others returned at once): wg_rel = first channel of the workgroup, relative
template <int CPL, class ARGS>
__device__ __forceinline__ int mixbus_live_waves(const ARGS &a, size_t wg_rel, int waves) {
    const size_t left = a.n_launch > wg_rel ? a.n_launch - wg_rel : 0;
    const size_t n = (left + 64 * CPL - 1) / (64 * CPL);
    return n < (size_t)waves ? (int)n : waves;
}
// after chunk [f0, f0 + F): flush the segment it completed, if any (uniform over the workgroup).  Everything the flush
// needs is derived here from the block / thread index rather than carried through the chunk loop in registers.
template <int F, int CPL>
__device__ __forceinline__ void mixbus_after_chunk(const ChainArgs &hot, MixStage &ms, unsigned f0) {
    const unsigned end = f0 + F;
    if (end % MIX_SEG != 0 && end != hot.nframes) return;
    const ColdArgs a = cold_args();
    const unsigned len = end % MIX_SEG ? end % MIX_SEG : MIX_SEG;
    const unsigned wb = blockDim.x == WG ? work_block(a.xcd_remap) : blockIdx.x;   // guarded tail launches use 64-lane blocks
    const int nw = mixbus_live_waves<CPL>(a, (size_t)wb * blockDim.x * CPL, (int)(blockDim.x / 64));
    const bool per_wave = a.mix_per_wave && blockDim.x == WG;
    mixbus_flush(a, ms, end - len, len, a.wave_base + (per_wave ? wb * (WG / 64) : wb), nw, __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), threadIdx.x & 63, per_wave);
}
// The same-block bus' tail for the rows this workgroup wrote (called by every wave at the end of a chain / graph kernel).
__device__ __forceinline__ void mix_tail_rows(unsigned wb, int wave, int lane) {
    const ColdArgs a = cold_args();
    if (!a.mt_tickets) return;
    if (a.mix_per_wave && blockDim.x == WG) mix_tail(a.wave_base + wb * (WG / 64) + (unsigned)wave, lane);   // (waves past n_launch have returned)
    else if (wave == 0) mix_tail(a.wave_base + wb, lane);
}

template <int F, int CPL, class SL, bool MOD = false>
__device__ __forceinline__ void chain_chunk(const ChainArgs &a, float (&st)[MAX_SLOTS][4][CPL], size_t c, const WaveAddr &w,
                                            unsigned f0, int lane, MixStage &ms, int wave) {
    float v[F][CPL];
#pragma unroll
    for (int f = 0; f < F; ++f) load_row<CPL, S_IN>(row_rsrc(a.in + w.io_base0 + (size_t)f0 * a.ld), w.io_off, (unsigned)f * (a.ld * 4u), v[f]);
    const Ctx cx{c, a.N, w.io_base0, w.io_off, w.ring_base0, w.ring_off, a.ld, f0, a.hop_div, a.hop_rc, a.third_rc, a.side, a.side_hop, true};
    // delay taps first (see RingPre), then the nodes in order
    RingPre<F, CPL> pre[MAX_SLOTS];
#define DSPFX_PF(I) if constexpr (sig_is<K_REVERB>(SL::v[I])) ring_prefetch<F, CPL, false>(a.slot[I], cx, pre[I]);
    DSPFX_FOR_SLOTS(DSPFX_PF)
#undef DSPFX_PF
#define DSPFX_RUN(I)                                                                 \
    if constexpr (sig_is<K_REVERB>(SL::v[I])) {                                      \
        if constexpr (sig_hop(SL::v[I])) apply_hop<F, CPL, true>(v, cx.hop_div, cx.hop_rc); \
        ring_apply<F, CPL, false>(a.slot[I], v, pre[I], cx);                         \
    } else {                                                                         \
        run_slot<SL::v[I], F, CPL, false, true, MOD>(a.slot[I], v, st[I], cx);       \
    }
    DSPFX_FOR_SLOTS(DSPFX_RUN)
#undef DSPFX_RUN
#pragma unroll
    for (int f = 0; f < F; ++f)
        if (!a.skip_store) store_row<CPL, S_OUT>(row_rsrc(a.out + w.io_base0 + (size_t)f0 * a.ld), w.io_off, (unsigned)f * (a.ld * 4u), v[f]);
    if (a.mixpart) mixbus_partial<F, CPL>(ms, v, true, f0, lane, wave);
}

// DSPFX_TS_TRACE (tools/wg_timeline.py, a debug build only): per wave of the last launch, wall-clock stamps (100 MHz) at entry and
// after its last store has been acknowledged, and where it ran (HW_ID: SIMD, CU, shader engine; XCC_ID)
#ifdef DSPFX_TS_TRACE
constexpr unsigned WG_TRACE_WAVES = 32768;
static __device__ unsigned long long dspfx_wg_trace[WG_TRACE_WAVES * 3];
#define DSPFX_WG_IN const unsigned long long wg_t_in = wall_clock64();
#define DSPFX_WG_OUT(WB, WAVE)                                                                                      \
    {                                                                                                               \
        __builtin_amdgcn_s_waitcnt(0);                                                                              \
        const unsigned wi = (WB) * (WG / 64) + (WAVE);                                                              \
        if (lane == 0 && wi < WG_TRACE_WAVES) {                                                                     \
            dspfx_wg_trace[wi * 3] = wg_t_in;                                                                       \
            dspfx_wg_trace[wi * 3 + 1] = wall_clock64();                                                            \
            dspfx_wg_trace[wi * 3 + 2] = ((unsigned long long)__builtin_amdgcn_s_getreg((31 << 11) | 20) << 32) |  \
                                         (unsigned)__builtin_amdgcn_s_getreg((31 << 11) | 4);                      \
        }                                                                                                           \
    }
#else
#define DSPFX_WG_IN
#define DSPFX_WG_OUT(WB, WAVE)
#endif
// Covers channels a.c_base + [0, a.n_launch), n_launch % (64*CPL) == 0 (whole waves).
template <int F, int CPL, class SL, bool MOD = false>
__global__ void __launch_bounds__(WG) chain_kernel(const ChainArgs a) {
    DSPFX_WG_IN
    __shared__ MixStage ms;
    if (a.mp_stage) mixpipe_prologue(a);
    const unsigned wb = work_block(a.xcd_remap);
    const unsigned tid = wb * WG + threadIdx.x;
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const size_t rel = (size_t)tid * CPL;
    if (rel >= a.n_launch) return;                 // whole-wave uniform by construction (a finished wave holds up no barrier)
    const size_t c = a.c_base + rel;
    float st[MAX_SLOTS][4][CPL];
#define DSPFX_LD(I) load_state<SL::v[I], CPL, false>(a.slot[I], st[I], c, a.N, true);
    DSPFX_FOR_SLOTS(DSPFX_LD)
#undef DSPFX_LD
    const WaveAddr w = wave_addr(a, c);
    unsigned f0 = 0;
    for (; f0 + F <= a.nframes; f0 += F) {
        chain_chunk<F, CPL, SL, MOD>(a, st, c, w, f0, lane, ms, wave);
        if (a.mixpart) mixbus_after_chunk<F, CPL>(a, ms, f0);
    }
    if constexpr (F > 1)
        for (; f0 < a.nframes; ++f0) {
            chain_chunk<1, CPL, SL, MOD>(a, st, c, w, f0, lane, ms, wave);
            if (a.mixpart) mixbus_after_chunk<1, CPL>(a, ms, f0);
        }
#define DSPFX_ST(I)                                                                              \
    if constexpr (sig_is<K_SIGNAL_GEN>(SL::v[I])) signal_gen_close_block<CPL>(a.slot[I], st[I], a.nframes); \
    store_state<SL::v[I], CPL, false>(a.slot[I], st[I], c, a.N, true);
    DSPFX_FOR_SLOTS(DSPFX_ST)
#undef DSPFX_ST
    DSPFX_WG_OUT(wb, wave)
    if (a.mt_tickets) mix_tail_rows(work_block(a.xcd_remap), wave, lane);
}

// ---- the fused chain kernel, time-sliced: few channels ------------------------------------------------
// With <= 2 waves per SIMD (N <= 131072 at one channel per lane) nothing hides a wave's memory and instruction
// latency: the 65536-channel 3-node chain ran at 0.49 of the HBM peak (profiles/r01_small_n.txt).  Here a workgroup
// owns 64*CPL channels and its four waves each take S = 32 of the block's 128 frames.  Every wave issues ALL its loads
// (samples and delay taps) at once and walks the chain node by node on its own slice: stateless nodes and delay lines
// (whose taps never depend on the block itself) run in all four waves at the same time; a node with per-channel state
// (biquad, one-pole, envelope, generator) is a recurrence over time, so the waves take turns at it -- wave q starts from
// the state wave q-1 left in LDS -- and every recurrence still sees its frames in order: bit for bit chain_kernel's
// results, with four times the waves per SIMD.
// DSPFX_TS_TRACE (tools/ts_timeline.py, a debug build only): wall-clock stamps (100 MHz) per wave at the phase boundaries
#ifdef DSPFX_TS_TRACE
constexpr unsigned TS_TRACE_GROUPS = 4096, TS_TRACE_STAMPS = 16;
static __device__ unsigned long long dspfx_ts_trace[TS_TRACE_GROUPS * 4 * TS_TRACE_STAMPS];
#define DSPFX_TS_STAMP(k)                                                                                          \
    if (lane == 0 && group < TS_TRACE_GROUPS) dspfx_ts_trace[((size_t)group * 4 + q) * TS_TRACE_STAMPS + (k)] = wall_clock64();
#else
#define DSPFX_TS_STAMP(k)
#endif
// first slot with per-channel state (a recurrence: turns) / first delay slot of a chain shape, or MAX_SLOTS
template <class SL> constexpr int ts_first_stateful() {
    for (int i = 0; i < MAX_SLOTS; ++i)
        if (SL::v[i] != SIG_NONE && !sig_is<K_REVERB>(SL::v[i]) && kind_nstate(sig_kind(SL::v[i])) > 0) return i;
    return MAX_SLOTS;
}
template <class SL> constexpr int ts_slot_count() {
    int n = 0;
    for (int i = 0; i < MAX_SLOTS; ++i) n += SL::v[i] != SIG_NONE ? 1 : 0;
    return n;
}
template <class SL> constexpr int ts_first_reverb() {
    for (int i = 0; i < MAX_SLOTS; ++i)
        if (sig_is<K_REVERB>(SL::v[i])) return i;
    return MAX_SLOTS;
}
// The chain on one wave's slice.  LATE (slice 0 only): the delay taps are requested AFTER the wave's turn at the first stateful
// node instead of right behind the block's samples.  A wave issues in order, and under the memory system's back-pressure
// the 32 tap loads take microseconds to ISSUE: behind them slice 0 could not start the chain of turns -- which every other
// slice waits for -- before the whole workgroup's reads were served.  The other slices request their taps first: they
// have to wait for their turn anyway.  (Two copies of the body rather than one with conditional loads: at a join of
// paths with different loads in flight the compiler's counter pass waits for all of them.)
template <int S, int CPL, class SL, bool LATE, bool GUARD>
__device__ __forceinline__ void ts_run(const ChainArgs &a, float (&lds_st)[4][CPL][64], const Ctx &cx, const WaveAddr &w, float (&v)[S][CPL],
                                       int q, int lane, size_t c, unsigned group, MixStage &ms, unsigned f_begin, bool active) {
    constexpr int FS = ts_first_stateful<SL>(), FR = ts_first_reverb<SL>();
    constexpr bool late = LATE && FS < FR && FR < MAX_SLOTS;
    // rows through buffer descriptors (load_row / store_row): the lane's byte offsets, out of range for lanes past the last channel
    const unsigned io_voff = (GUARD && !active) ? ROW_OOB : w.io_off, ring_voff = (GUARD && !active) ? ROW_OOB : w.ring_off;
    const unsigned row_bytes = a.ld * 4u;
    const __amdgpu_buffer_rsrc_t r_out = row_rsrc(a.out + w.io_base0 + (size_t)f_begin * a.ld);
    // delay taps of every delay node of the chain (nframes <= D: they never depend on this block's outputs)
#define DSPFX_TAPS_DECL(I)                                                                                       \
    float tap##I[sig_is<K_REVERB>(SL::v[I]) ? S : 1][CPL];                                                       \
    RingGroups rg##I{};        /* three pointers: the 32 row addresses are formed again for the stores */
    DSPFX_FOR_SLOTS(DSPFX_TAPS_DECL)
#undef DSPFX_TAPS_DECL
#define DSPFX_TAPS(I)                                                                                            \
    if constexpr (sig_is<K_REVERB>(SL::v[I])) {                                                                  \
        rg##I = ring_groups_host(a.slot[I]);   /* blocks of exactly 128 frames: the host named the groups (no table read) */ \
        const unsigned pos_ld = opaque_u32(a.slot[I].pos + cx.f0);   /* formed HERE, not at the top of the kernel */  \
        _Pragma("unroll") for (int f = 0; f < S; ++f)                                                            \
        {                                                                                                        \
            const RowRef rr = ring_row_ref(a.slot[I], cx, rg##I, pos_ld, f);                                         \
            load_row<CPL, S_RING_LD>(rr.rsrc, ring_voff, rr.soff, tap##I[f]);                                        \
            __builtin_amdgcn_sched_barrier(0);   /* one row's scalar operands at a time */                          \
        }                                                                                                        \
    }
    auto load_taps = [&]() __attribute__((always_inline)) { DSPFX_FOR_SLOTS(DSPFX_TAPS) };
    // Slice 0's own copy of the body (LATE) also requests the state of the first stateful node right here, behind its samples:
    // loaded inside its turn it was one more memory round trip on the chain of turns that every other slice waits for.
    float st_pre[4][CPL];
    if constexpr (LATE && FS < MAX_SLOTS) load_state<SL::v[FS < MAX_SLOTS ? FS : 0], CPL, GUARD>(a.slot[FS < MAX_SLOTS ? FS : 0], st_pre, c, a.N, active);
    if constexpr (!late) load_taps();
    DSPFX_TS_STAMP(1)
#define DSPFX_RUN(I)                                                                                             \
    if constexpr (SL::v[I] != SIG_NONE) { DSPFX_TS_STAMP(2 + I) }                                                \
    if constexpr (sig_is<K_REVERB>(SL::v[I])) {                                                                  \
        if constexpr (sig_hop(SL::v[I])) apply_hop<S, CPL, true>(v, cx.hop_div, cx.hop_rc);                      \
        const float decay = a.slot[I].p[0];                                                                      \
        const unsigned pos_st = opaque_u32(a.slot[I].pos + cx.f0);                                               \
        ring_zero_cleared<sig_is<K_REVERB>(SL::v[I]) ? S : 1, CPL>(a.slot[I], cx, tap##I);   /* taps from before the ring's last clear: +0.0 */ \
        _Pragma("unroll") for (int f = 0; f < S; ++f) {                                                          \
            _Pragma("unroll") for (int j = 0; j < CPL; ++j) v[f][j] = v[f][j] + tap##I[f][j] * decay;            \
            const RowRef rr = ring_row_ref(a.slot[I], cx, rg##I, pos_st, f);                                         \
            store_row<CPL, S_RING_ST>(rr.rsrc, ring_voff, rr.soff, v[f]);                                            \
            __builtin_amdgcn_sched_barrier(0);                                                                   \
        }                                                                                                        \
    } else if constexpr (SL::v[I] != SIG_NONE && kind_nstate(sig_kind(SL::v[I])) == 0) {                         \
        float none[4][CPL];                                                                                      \
        run_slot<SL::v[I], S, CPL, GUARD, true, false>(a.slot[I], v, none, cx);                                  \
    } else if constexpr (SL::v[I] != SIG_NONE) {                                                                 \
        constexpr int NS = kind_nstate(sig_kind(SL::v[I]));                                                      \
        _Pragma("unroll 1") for (int turn = 0; turn < 4; ++turn) {                                               \
            if (q == turn) {                                                                                     \
                float st[4][CPL];                                                                                \
                if (turn == 0) {                                                                                 \
                    if constexpr (LATE && I == FS) {                                                             \
                        _Pragma("unroll") for (int k = 0; k < 4; ++k)                                            \
                            _Pragma("unroll") for (int j = 0; j < CPL; ++j) st[k][j] = st_pre[k][j];             \
                    } else load_state<SL::v[I], CPL, GUARD>(a.slot[I], st, c, a.N, active);                        \
                } else {                                                                                           \
                    _Pragma("unroll") for (int k = 0; k < 4; ++k)                                                \
                        _Pragma("unroll") for (int j = 0; j < CPL; ++j) st[k][j] = k < NS ? lds_st[k][j][lane] : 0.0f; \
                }                                                                                                \
                run_slot<SL::v[I], S, CPL, GUARD, true, false>(a.slot[I], v, st, cx);                            \
                if (turn == 3) {                                                                                 \
                    if constexpr (sig_is<K_SIGNAL_GEN>(SL::v[I])) signal_gen_close_block<CPL>(a.slot[I], st, a.nframes); \
                    store_state<SL::v[I], CPL, GUARD>(a.slot[I], st, c, a.N, active);                              \
                } else {                                                                                         \
                    _Pragma("unroll") for (int k = 0; k < NS; ++k)                                               \
                        _Pragma("unroll") for (int j = 0; j < CPL; ++j) lds_st[k][j][lane] = st[k][j];           \
                }                                                                                                \
                if constexpr (late && I == FS) load_taps();      /* slice 0, its first turn done */             \
            }                                                                                                    \
            __syncthreads();                                                                                     \
        }                                                                                                        \
    }
    DSPFX_FOR_SLOTS(DSPFX_RUN)
#undef DSPFX_RUN
#undef DSPFX_TAPS
    DSPFX_TS_STAMP(12)
#pragma unroll
    for (int f = 0; f < S; ++f)
        if (!a.skip_store) {
            store_row<CPL, S_OUT>(r_out, io_voff, (unsigned)f * row_bytes, v[f]);
            __builtin_amdgcn_sched_barrier(0);
        }
    DSPFX_TS_STAMP(13)
    if (a.mixpart) mixbus_partial<S, CPL>(ms, v, active, f_begin, lane, 0);      // the four slices of ONE row
    DSPFX_TS_STAMP(14)
}

// GUARD: the launch for the N % (64 * CPL) channels the whole-wave launch leaves over (one workgroup per 64 channels, CPL = 1):
// out-of-range lanes stay alive, read zeros and store nothing -- four slices in parallel instead of the interpreter's one wave
// walking the block chunk by chunk (43 -> ~9 us behind the main launch).
template <int S, int CPL, class SL, bool GUARD = false>
__global__ void __launch_bounds__(WG) chain_ts_kernel(const ChainArgs a) {
    // (Passing the state from slice to slice with an LDS flag instead of a workgroup barrier per turn -- so that the early
    // slices store while the later ones still take their turns -- was tried: config 2 27.3 -> 30.2 us, the 5-node chain at
    // 65536 channels 35.0 -> 32.7 us only at two channels per lane; reads and writes in separate phases suit the HBM better.)
    __shared__ float lds_st[4][CPL][64];           // the state rows of the node whose turns are being taken
    __shared__ MixStage ms;
    if (a.mp_stage) mixpipe_prologue(a);
    const int lane = threadIdx.x & 63;
    const int q = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);      // time slice of this wave
    // (Rotating the slices over the waves per workgroup -- so that co-resident workgroups take their turns on different
    // SIMDs -- changes nothing: profiles/r02_small_n.txt.)
    const unsigned group = work_block(a.xcd_remap);                        // channel group of 64*CPL channels
    const unsigned wave_global = a.wave_base + group;
    const size_t rel = ((size_t)group * 64 + lane) * CPL;
    bool active = true;
    if constexpr (GUARD) active = rel < a.n_launch;
    else if (rel >= a.n_launch) return;            // uniform over the whole workgroup: no barrier is left waiting
    DSPFX_TS_STAMP(0)
    const size_t c = a.c_base + (active ? rel : 0);
    const WaveAddr w = wave_addr(a, c);
    const unsigned f_begin = (unsigned)q * S;
    const Ctx cx{c, a.N, w.io_base0, w.io_off, w.ring_base0, w.ring_off, a.ld, f_begin, a.hop_div, a.hop_rc, a.third_rc, a.side, a.side_hop, active};
    float v[S][CPL];
    {
        const unsigned io_voff = (GUARD && !active) ? ROW_OOB : w.io_off, row_bytes = a.ld * 4u;
        const __amdgpu_buffer_rsrc_t r_in = row_rsrc(a.in + w.io_base0 + (size_t)f_begin * a.ld);
#pragma unroll
        for (int f = 0; f < S; ++f) {
            load_row<CPL, S_IN>(r_in, io_voff, (unsigned)f * row_bytes, v[f]);
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    // slice 0's late taps cost registers (the second copy of the body): 3-node chain 119 -> 153 (one channel per lane, still
    // three workgroups per CU where that form is used), 218 -> 243 (two); the 5-node chain would drop from four workgroups
    // per CU to three (108 -> 155) and ran 36 -> 42 us at 65536 channels: short chains only.  (Late taps for EVERY slice --
    // one copy of the body -- expose the taps' latency in the later slices: config 2 33.4 us.)
    if constexpr (ts_slot_count<SL>() <= 3) {
        if (q == 0) ts_run<S, CPL, SL, true, GUARD>(a, lds_st, cx, w, v, q, lane, c, group, ms, f_begin, active);
        else ts_run<S, CPL, SL, false, GUARD>(a, lds_st, cx, w, v, q, lane, c, group, ms, f_begin, active);
    } else {
        ts_run<S, CPL, SL, false, GUARD>(a, lds_st, cx, w, v, q, lane, c, group, ms, f_begin, active);
    }
    if (a.mixpart) {                               // 4 S frames == one segment; the workgroup's row is the four slices side by side
        mixbus_flush(a, ms, 0, 4 * S, wave_global, 1, q, lane);
        if (a.mt_tickets && q == 0) mix_tail(wave_global, lane);
    }
}

// ---- the fused chain kernel, interpreting any chain ----------------------------------
// One lane = one channel.  The node loop runs at run time (wave-uniform scalar
// branches); per-node filter state is staged in LDS ([row][lane], conflict-free)
// for the whole block and touched once per chunk per stateful node.
// GUARD=true: one-wave tail launch whose out-of-range lanes stay alive with zeros.
// MOD=true additionally evaluates connected / latched `as_input` sliders (control ports); it is a
// separate instantiation because those paths nearly double the register footprint (it runs at F=4).
template <int F, int CPL, bool GUARD, bool FAST, bool MOD, bool LIBM>
__device__ __forceinline__ void dyn_chunk(const ChainArgs &a, float *lds, size_t c, const WaveAddr &w, bool active, unsigned f0,
                                          int lane, MixStage &ms, int wave) {
    float v[F][CPL];
#pragma unroll
    for (int f = 0; f < F; ++f)
        load_row<CPL, S_IN>(row_rsrc(a.in + w.io_base0 + (size_t)f0 * a.ld), (GUARD && !active) ? ROW_OOB : w.io_off, (unsigned)f * (a.ld * 4u), v[f]);
    const Ctx cx{c, a.N, w.io_base0, w.io_off, w.ring_base0, w.ring_off, a.ld, f0, a.hop_div, a.hop_rc, a.third_rc, a.side, a.side_hop, active};
    // (Prefetching the first delay node's taps here, as the static kernels do, was measured: +18 VGPRs cost a wave
    // of occupancy and the 5-node chain went from 0.4255 to 0.4565 ms.  The interpreter loads them in the node's slot.)
    int row = 0;
#pragma unroll 1
    for (int s = 0; s < a.n_slots; ++s) {
        const SlotArgs sl = a.slot[s];   // a private copy keeps the kernarg table out of scratch
        const int ns = slot_nstate<SIG_DYN>(sl);
        float st[4][CPL];
#pragma unroll
        for (int k = 0; k < 4; ++k)
#pragma unroll
            for (int j = 0; j < CPL; ++j) st[k][j] = (k < ns) ? lds[((row + k) * CPL + j) * WG + threadIdx.x] : 0.0f;
        run_slot<SIG_DYN, F, CPL, GUARD, FAST, MOD, LIBM>(sl, v, st, cx);
#pragma unroll
        for (int k = 0; k < 4; ++k)
            if (k < ns) {
#pragma unroll
                for (int j = 0; j < CPL; ++j) lds[((row + k) * CPL + j) * WG + threadIdx.x] = st[k][j];
            }
        row += ns;
    }
#pragma unroll
    for (int f = 0; f < F; ++f)
        if (!a.skip_store)
            store_row<CPL, S_OUT>(row_rsrc(a.out + w.io_base0 + (size_t)f0 * a.ld), (GUARD && !active) ? ROW_OOB : w.io_off, (unsigned)f * (a.ld * 4u), v[f]);
    if (a.mixpart) mixbus_partial<F, CPL>(ms, v, !GUARD || active, f0, lane, wave);
}

// LDS: [state rows][CPL][WG] floats (conflict-free: consecutive lanes, consecutive banks)
template <int F, int CPL, bool GUARD, bool MOD, bool LIBM>
__global__ void __launch_bounds__(WG) chain_dyn_kernel(const ChainArgs a) {
    extern __shared__ float lds[];
    __shared__ MixStage ms;
    if (a.mp_stage) mixpipe_prologue(a);
    const unsigned wb = blockDim.x == WG ? work_block(a.xcd_remap) : blockIdx.x;   // tail launches use 64-lane blocks
    const unsigned tid = wb * blockDim.x + threadIdx.x;
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const size_t rel = (size_t)tid * CPL;
    const bool active = rel < a.n_launch;
    if (!GUARD && !active) return;                 // whole-wave uniform by construction
    const size_t c = a.c_base + (active ? rel : 0);
    {
        int row = 0;
#pragma unroll 1
        for (int s = 0; s < a.n_slots; ++s) {
            const int ns = slot_nstate<SIG_DYN>(a.slot[s]), np = slot_npersist<SIG_DYN>(a.slot[s]);
            for (int k = 0; k < ns; ++k) {
                float t[CPL];
#pragma unroll
                for (int j = 0; j < CPL; ++j) t[j] = 0.0f;
                if (k < np) load_vec<CPL, GUARD>(a.slot[s].state + (size_t)k * a.N + c, t, active);
#pragma unroll
                for (int j = 0; j < CPL; ++j) lds[((row + k) * CPL + j) * WG + threadIdx.x] = t[j];
            }
            row += ns;
        }
    }
    const WaveAddr w = wave_addr(a, c);
    unsigned f0 = 0;
    if (a.fast_div) {
        for (; f0 + F <= a.nframes; f0 += F) {
            dyn_chunk<F, CPL, GUARD, true, MOD, LIBM>(a, lds, c, w, active, f0, lane, ms, wave);
            if (a.mixpart) mixbus_after_chunk<F, CPL>(a, ms, f0);
        }
        if constexpr (F > 1)
            for (; f0 < a.nframes; ++f0) {
                dyn_chunk<1, CPL, GUARD, true, MOD, LIBM>(a, lds, c, w, active, f0, lane, ms, wave);
                if (a.mixpart) mixbus_after_chunk<1, CPL>(a, ms, f0);
            }
    } else {
        for (; f0 + F <= a.nframes; f0 += F) {
            dyn_chunk<F, CPL, GUARD, false, MOD, LIBM>(a, lds, c, w, active, f0, lane, ms, wave);
            if (a.mixpart) mixbus_after_chunk<F, CPL>(a, ms, f0);
        }
        if constexpr (F > 1)
            for (; f0 < a.nframes; ++f0) {
                dyn_chunk<1, CPL, GUARD, false, MOD, LIBM>(a, lds, c, w, active, f0, lane, ms, wave);
                if (a.mixpart) mixbus_after_chunk<1, CPL>(a, ms, f0);
            }
    }
    {
        int row = 0;
#pragma unroll 1
        for (int s = 0; s < a.n_slots; ++s) {
            const int ns = slot_nstate<SIG_DYN>(a.slot[s]), np = slot_npersist<SIG_DYN>(a.slot[s]);
            if (a.slot[s].kind == K_SIGNAL_GEN) {
                float st2[4][CPL];
#pragma unroll
                for (int j = 0; j < CPL; ++j) {
                    st2[0][j] = lds[((row + 0) * CPL + j) * WG + threadIdx.x];
                    st2[1][j] = lds[((row + 1) * CPL + j) * WG + threadIdx.x];
                    st2[2][j] = 0.0f;
                    st2[3][j] = 0.0f;
                }
                signal_gen_close_block<CPL>(a.slot[s], st2, a.nframes);
#pragma unroll
                for (int j = 0; j < CPL; ++j) lds[((row + 0) * CPL + j) * WG + threadIdx.x] = st2[0][j];
            }
            for (int k = 0; k < np; ++k) {
                float t[CPL];
#pragma unroll
                for (int j = 0; j < CPL; ++j) t[j] = lds[((row + k) * CPL + j) * WG + threadIdx.x];
                store_vec<CPL, GUARD>(a.slot[s].state + (size_t)k * a.N + c, t, active);
            }
            row += ns;
        }
    }
    if (a.mt_tickets) mix_tail_rows(wb, wave, lane);
}

}  // namespace dspfx
