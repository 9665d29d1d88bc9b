/*
 * dspfx.h -- C ABI of the MI355X-native effect-chain engine.
 *
 * The drop-in boundary for simmsb/dsp-stuff's per-block effect-node evaluation
 * loop: everything `SimpleNode::process` (dsp-stuff/src/node.rs:135-146) and the
 * blanket `Perform` wrapper (node.rs:267-352) compute for one mono channel and one
 * 128-frame block, evaluated here for N independent channels per launch by
 * hand-written HIP kernels (gfx950).  The reference has no FFI of its own
 * (SURVEY.md 8b); this header is the ABI its Rust host would bind (see
 * INTEGRATION.md for the `extern "C"` block and the `GpuChain: SimpleNode` shim).
 *
 * Conventions
 *   - plain C types only; every call returns 0 (DSPFX_OK) or a negative
 *     dspfx_status; nothing aborts or throws across the boundary (the reference
 *     panics instead: node.rs:173,271,280).
 *   - the caller owns sample buffers; the engine owns parameters, coefficients
 *     and all DSP state (biquad history, one-pole z, delay rings, FIR history)
 *     in HBM -- the same ownership split as node.rs:271-288 vs biquad.rs:43-44,
 *     reverb.rs:40-41, fir.rs:64-65.
 *   - sample layout is frame-major f32 by default: buf[frame * channels + channel]
 *     ("[B][N]"): for every frame the N channels are contiguous, so one
 *     wavefront reads 64 consecutive channels as one coalesced burst
 *     (dspfx_engine_desc.tile_channels selects the channel-tiled form).
 *   - threads: every entry point may be called from any thread; calls on one engine are
 *     serialised by the engine (distinct engines are independent).  The reference's split is
 *     kept: ONE thread drives the process calls (one node task, runtime.rs:718-728) while
 *     another -- the GUI -- stores sliders and modes (dsp-stuff-derive/src/lib.rs:487-492,
 *     biquad.rs:62-76).  dspfx_set_param / dspfx_set_mode NEVER wait for a process call in
 *     progress: the store is queued and takes effect at the next block boundary (at once when
 *     the engine is idle), in the order the stores were made.
 *   - streams: the process calls are asynchronous on the caller's stream.  Everything that
 *     writes DSP state outside a block (a biquad's reset on a coefficient store, dspfx_reset)
 *     is queued on the stream the state was last used on, so it is ordered behind the blocks
 *     in flight and ahead of the next one -- with any kind of stream (torch's and most hosts'
 *     streams are non-blocking: the null stream orders nothing against them).  A process call
 *     on a DIFFERENT stream first waits (on the device) for the previous stream.  Calls that
 *     free or re-allocate state (dspfx_chain_set, dspfx_graph_set, dspfx_set_taps, dspfx_ring_trim,
 *     dspfx_state_import / _export) wait for the device first.  A delay-ring length change does NOT (dspfx_set_param).
 *   - there is NO CPU fallback: without a HIP device every entry point that
 *     needs one fails with DSPFX_ERR_NO_DEVICE.
 *   - kernels: which kernel serves a chain is the engine's business and never changes a sample
 *     (dspfx_describe names it).  Kernels specialised for a chain's shape are taken from the
 *     on-disk cache or compiled by the library's background thread while an interpreting kernel
 *     serves, and adopted at a block boundary (dspfx_kernels_ready).  N need not be a multiple of 64.
 */
#ifndef DSPFX_H
#define DSPFX_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* 2 (round 4): REVERB carries its `seconds` slider (params[1]) and every slider store on a REVERB node swaps in a new zero
 *    ring like the reference's after_settings_change; the FIR state blob is {32-byte header, held samples} (round 3);
 *    DSPFX_FIR_PRECISION_DEFAULT means the split bf16 sweep (round 3).  Engines refuse a descriptor of another version. */
#define DSPFX_ABI_VERSION 2
/* dsp-stuff/src/node.rs:257 `pub const BUF_SIZE: usize = 128;` */
#define DSPFX_BUF_SIZE 128
/* longest chain one engine accepts */
#define DSPFX_MAX_NODES 32

typedef struct dspfx_engine dspfx_engine;

typedef enum dspfx_status {
    DSPFX_OK = 0,
    DSPFX_ERR_INVALID = -1,    /* bad argument / descriptor */
    DSPFX_ERR_NO_DEVICE = -2,  /* no usable HIP device (there is no CPU fallback) */
    DSPFX_ERR_HIP = -3,        /* a HIP runtime call failed; see dspfx_last_error */
    DSPFX_ERR_OOM = -4,        /* device allocation failed */
    DSPFX_ERR_UNSUPPORTED = -5,
    DSPFX_ERR_STATE = -6       /* call not valid in the engine's current state */
} dspfx_status;

/* Node kinds = the effect variants of `enum Nodes` (nodes/mod.rs:38-63) that
 * are on the hot path. */
typedef enum dspfx_kind {
    DSPFX_GAIN = 0,      /* nodes/gain.rs:25-38 */
    DSPFX_BIQUAD = 1,    /* nodes/biquad.rs:48-88 (+ biquad 0.4.2 DirectForm1<f32>) */
    DSPFX_LOW_PASS = 2,  /* nodes/low_pass.rs:26-42 */
    DSPFX_HIGH_PASS = 3, /* nodes/high_pass.rs:26-42 */
    DSPFX_REVERB = 4,    /* nodes/reverb.rs:44-111: feedback delay line */
    DSPFX_DISTORT = 5,   /* nodes/distort.rs:53-195 */
    DSPFX_OVERDRIVE = 6, /* nodes/overdrive.rs:31-72 */
    DSPFX_CHEBYSHEV = 7, /* nodes/chebyshev.rs:28-62 */
    DSPFX_FIR = 8,       /* nodes/fir.rs:179-225 */
    DSPFX_ADD = 9,       /* nodes/add.rs:24-34  (port "b" = the side input) */
    DSPFX_MIX = 10,      /* nodes/mix.rs:31-47  (port "b" = the side input) */
    DSPFX_SIGNAL_GEN = 11, /* nodes/signal_gen.rs:55-129: a SOURCE (no "in" port): replaces the signal */
    DSPFX_ENVELOPE = 12, /* nodes/envelope.rs:34-52: dasp_envelope 0.11.0 peak detector (full-wave) */
    DSPFX_N_KINDS = 13
} dspfx_kind;

/* nodes/signal_gen.rs:17-22 `enum Mode` */
typedef enum dspfx_signal_mode { DSPFX_SIG_SINE = 0, DSPFX_SIG_TRIANGLE = 1, DSPFX_SIG_SQUARE = 2, DSPFX_SIG_CONSTANT = 3 } dspfx_signal_mode;

/* nodes/distort.rs:18-28 `enum Mode`, declaration order (repr(u8)) */
typedef enum dspfx_distort_mode {
    DSPFX_DIST_HARD_CLIP = 0,
    DSPFX_DIST_SOFT_CLIP = 1,
    DSPFX_DIST_TANH = 2,
    DSPFX_DIST_RECIP_SOFT_CLIP = 3,
    DSPFX_DIST_FUZZ = 4,
    DSPFX_DIST_SIN = 5,
    DSPFX_DIST_ATAN = 6,
    DSPFX_DIST_SQUARE = 7,
    DSPFX_DIST_CHEBYSHEV4 = 8
} dspfx_distort_mode;

/* nodes/fir.rs `enum Mode` (fir.rs:187-190) */
typedef enum dspfx_fir_mode { DSPFX_FIR_BALANCED = 0, DSPFX_FIR_AVERAGE = 1 } dspfx_fir_mode;

/* Which hops of the chain reproduce `collect_and_average` with one connected
 * pipe (node.rs:162-194: value = (0.0 + x) / f32(0.0001 + 1.0)).
 *   INTERNAL: the hops between consecutive nodes of the chain (what disappears
 *             when k reference nodes are fused into one GpuChain node);
 *   INPUT   : also the hop into the first node (whole-graph semantics: the
 *             engine is fed by the Input node, nodes/input.rs:213-240).        */
#define DSPFX_LINK_INTERNAL 1u
#define DSPFX_LINK_INPUT 2u
/*   SIDE_RAW: the side input (port "b" of ADD/MIX) is taken as given: it was already averaged
 *             over several links by dspfx_link_average (graphs with fan-in on that port).  */
#define DSPFX_LINK_SIDE_RAW 4u
/* most links into one port that dspfx_link_average accepts */
#define DSPFX_MAX_LINKS 16

typedef struct dspfx_engine_desc {
    uint32_t abi_version;     /* DSPFX_ABI_VERSION */
    int32_t device;           /* HIP device ordinal */
    uint32_t channels;        /* N: independent mono channels held by this engine */
    uint32_t max_frames;      /* largest n_frames a process call will pass (>=1) */
    uint32_t link_flags;      /* DSPFX_LINK_* */
    uint32_t tile_channels;   /* sample layout: 0 = frame-major [n_frames][N];
                                 W (power of two, N % W == 0) = channel-tiled [N/W][n_frames][W]:
                                 element (f, c) at ((c / W) * n_frames + f) * W + c % W, so the
                                 block of every W-channel group is one contiguous HBM extent */
    uint64_t channel_offset;  /* global index of local channel 0 (multi-GPU shards, noise) */
} dspfx_engine_desc;

/* One node of the chain: the reference node's slider fields, in field order.
 *   GAIN       params[0]=level (0..=10, default 1)                    gain.rs:21-22
 *   BIQUAD     params[0..5]=a0,a1,a2,b0,b1,b2 (raw sliders -10..=10)  biquad.rs:18-41
 *   LOW_PASS   params[0]=ratio (0..=1, default 0.5)                   low_pass.rs:20-21
 *   HIGH_PASS  params[0]=ratio                                        high_pass.rs:20-21
 *   REVERB     params[0]=decay (0..=1, default .5), params[1]=seconds (0..=1, default .5; 0 = not given);
 *              delay_len=D, the ring the node STARTS with (restored: dspfx_delay_len(seconds, r); fresh from the menu:
 *              dspfx_delay_len(0, r) -- make_buffer); mode bit 0 = r: the page-rounded reading of seconds -> samples
 *              (see dspfx_set_param for what a slider store does to the ring)   reverb.rs:29-38, 44-71
 *   DISTORT    params[0]=level (0..=30, default 0); mode              distort.rs:46-50
 *   OVERDRIVE  params[0]=boost, [1]=drive, [2]=level                  overdrive.rs:21-28
 *   CHEBYSHEV  params[0]=level_pos, [1]=level_neg                     chebyshev.rs:21-25
 *   FIR        taps/n_taps (time-REVERSED, as fir.rs:163,168 stores them); mode
 *   ADD        -
 *   MIX        params[0]=ratio (0..=1, default .5)                    mix.rs:22-28
 *   SIGNAL_GEN params[0]=amplitude (-1..=1, default .5), [1]=frequency (0.1..=20000 Hz, default 100); mode
 *              (dspfx_signal_mode); per-channel phase clock, wrapped at every 128-frame block end  signal_gen.rs:41-55
 *   ENVELOPE   params[0]=attack, [1]=release, both in frames (0..=1000, default 0); per-channel envelope
 *              env = d + (env - d)*g, d = |x|, g = env < d ? e^(-1/attack) : e^(-1/release), g = 0 for 0 frames
 *              (dasp_envelope 0.11.0 Detector::next, restated as recalled: see oracle/dspfx_oracle.h)  envelope.rs:27-30
 * delay_len is explicit because rivulet's capacity rounding is not in the
 * reference tree; dspfx_delay_len() gives both readings.    */
typedef struct dspfx_node_desc {
    int32_t kind;        /* dspfx_kind */
    int32_t mode;        /* dspfx_distort_mode / dspfx_fir_mode / dspfx_signal_mode */
    float params[8];
    uint32_t delay_len;  /* REVERB: D >= 128 */
    uint32_t n_taps;     /* FIR */
    const double *taps;  /* FIR: n_taps f64 values, host memory, copied */
} dspfx_node_desc;

/* ---- library ----------------------------------------------------------- */
uint32_t dspfx_abi_version(void);
const char *dspfx_strerror(int status);
/* Number of visible HIP devices (0 when there is none). */
int dspfx_device_count(void);
/* Fill `d` with the reference's defaults for `kind` (derive `default=`, lib.rs:196-210). */
int dspfx_node_defaults(int kind, dspfx_node_desc *d);
/* reverb.rs:58 `((seconds * 48000.0) as usize).max(128)`; page_round != 0 rounds up to whole 4 KiB pages (1024 f32).
 * Two readings of ONE fact that the reference tree does not contain (rivulet is a git dependency, Cargo.toml:42): both
 * refresh_seconds (reverb.rs:60-68) and make_buffer (reverb.rs:44-49) call circular_buffer::<f32>(n), try_grant(n) and then
 * release(view().len()) zeros -- the delay is however long the granted view is.
 *   page_round = 0  the view is exactly the n asked for: the delay is n samples (what the slider's "s" suffix promises);
 *   page_round = 1  the view is all the free space of a buffer whose capacity is whole pages: n rounded up to 1024 f32.
 * Evidence, as far as it goes: rivulet's circular buffer is a virtual-memory mirror (page-granular capacity) and its `grant`
 * contract is "at least count" -- both favour 1; the label "Delay ... s" and the 0.5 s default say what the author MEANT -- 0.
 * Neither can be pinned here, so the reading is the caller's choice (mode bit 0 of a REVERB node) and it is applied to BOTH
 * call sites alike: a node fresh from the menu sits on dspfx_delay_len(0, page_round) = 128 or 1024 samples. */
uint32_t dspfx_delay_len(float seconds, int page_round);
/* node.rs:166,179: f32 0.0001 incremented by 1.0 per connected pipe. */
float dspfx_link_divisor(uint64_t n_connected);

/* ---- engine lifecycle -------------------------------------------------- */
int dspfx_engine_create(const dspfx_engine_desc *desc, dspfx_engine **out);
/* Waits for the device (blocks in flight still read the engine's state), then frees everything.  Takes no lock: call it
 * when no other thread is inside an entry point of this engine or still holds a reference to it. */
void dspfx_engine_destroy(dspfx_engine *e);
const char *dspfx_last_error(const dspfx_engine *e);

/* Replace the chain (NodeStatic::new for every node, node.rs:125-133): allocates
 * and zeroes all DSP state.  Nodes are evaluated in order, node i feeding i+1
 * (a linear graph of runtime.rs LinkInstances). */
int dspfx_chain_set(dspfx_engine *e, const dspfx_node_desc *nodes, int n_nodes);
int dspfx_chain_len(const dspfx_engine *e);
/* Kernels specialised for a chain's shape come from this process' table or the on-disk cache of code objects
 * ($DSPFX_CACHE_DIR, else $XDG_CACHE_HOME/dspfx, else ~/.cache/dspfx; DSPFX_DISK_CACHE=0: none) in milliseconds; a shape seen
 * for the first time is compiled by ONE background thread while the engine serves its blocks on the interpreting kernel, and
 * adopted at a block boundary: dspfx_chain_set, dspfx_set_mode and the first connected control port never wait for the
 * compiler (the reference re-creates nodes on every graph edit, runtime.rs:319-362).  Samples are the same bit for bit, and so
 * is the bus (the interpreter runs with the coming kernel's rows of partial sums).  This call is for a host -- or a benchmark --
 * that wants the specialised kernels BEFORE its first block: it adopts what is finished and waits up to wait_ms milliseconds
 * for the rest.  Returns 1: nothing is pending (dspfx_describe names the kernels that are live), 0: still compiling, < 0: error.
 * It does not hold the engine while it waits.  DSPFX_JIT=0: no run-time kernels; DSPFX_JIT=1 / DSPFX_JIT_ASYNC=0: compiled inside
 * dspfx_chain_set (about a second per new shape), as in rounds 1-3.  A whole-graph kernel (dspfx_graph_set) has no interpreter to
 * stand in for it: compiled in the call when neither cache has it. */
int dspfx_kernels_ready(dspfx_engine *e, int wait_ms);

/* Slider store + `after_settings_change` (dsp-stuff-derive/src/lib.rs:487-497, 560-568: the generated render() runs the
 * node's hook when ANY of its widgets changed):
 *   BIQUAD renormalises by a0 and ZEROES its state (biquad.rs:15, 62-76);
 *   REVERB -- ANY slider, `decay` included -- swaps in a NEW ZERO-FILLED ring (reverb.rs:19, 55-71: refresh_seconds): the echo
 *          tail is cut.  Its length is what the seconds slider says, max((seconds * 48000) as usize, 128) (mode bit 0: rounded up
 *          to whole 4 KiB pages), when the node was given one (params[1] > 0) -- so a node fresh from the menu (dspfx_node_defaults:
 *          make_buffer's 128-sample ring -- 1024 page-rounded -- under a 0.5 s slider, reverb.rs:44-52) becomes a 24000-sample delay at its first slider
 *          change, like the reference's -- and the ring's current length otherwise.  The swap costs NOTHING on the thread
 *          that drives the blocks, whatever the two lengths: no wait for the device, no memset, no re-allocation, placement
 *          kept.  The ring is a table of separately allocated 128-row groups of which a ring of D samples uses the first
 *          ceil(D / 128); the swap sets D, restarts the position and makes the next D frames read their taps as +0.0 (a per-node
 *          frame counter in the kernel arguments), in order with the blocks in flight, which carry their own copies.  Groups a
 *          LONGER ring needs are allocated -- not zeroed: every row is written before it is read unmasked -- by the thread
 *          that MAKES the store, before the store is queued (the reference's GUI thread allocates the new ring too,
 *          reverb.rs:55-71); dspfx_chain_set reserves them up front for a menu-fresh node when that is cheap (at most 1/16 of the
 *          device's free memory; DSPFX_MENU_RING_RESERVE=0 never, =1 whenever it fits -- dspfx_reserve_delay_len is the explicit
 *          form and dspfx_ring_trim gives unused reservations back); a shorter ring keeps the surplus
 *          as capacity (dspfx_ring_trim returns it).  DSPFX_ERR_OOM: the store was NOT made, the node keeps ring and slider.
 *          params[1] outside 0..=1 (the slider's range, reverb.rs:34-37) is DSPFX_ERR_INVALID; a STORED 0.0 is a value like any other
 *          (a 128-sample ring, reverb.rs:58) -- only in a node DESCRIPTOR does params[1] = 0 mean "no seconds slider: keep delay_len";
 *   other kinds just take the value from the next block on.  Nothing is launched or compiled by a slider store (the
 * exactness of a DISTORT level as a constant divisor is decided on the host; only a level that is an even integer
 * other than a power of two runs the 2 ms device check, once per value and process).
 * Safe from a second thread while another one is inside a process call (the reference's GUI thread does exactly
 * that): the store is validated, queued and returns; it is applied -- with its stores before it, in order -- by the
 * next entry point that holds the engine: immediately when the engine is idle, else at the next block boundary.  The
 * biquad reset is queued on the stream of the blocks in flight, behind them.  dspfx_set_param_seq also returns the
 * store's sequence number; dspfx_param_log tells where each store took effect. */
int dspfx_set_param(dspfx_engine *e, int node, int param, float value);
int dspfx_set_param_seq(dspfx_engine *e, int node, int param, float value, uint64_t *seq);
/* Mode store (the `mode` atomics of distort.rs:46-50, fir.rs, signal_gen.rs): queued like a slider store. */
int dspfx_set_mode(dspfx_engine *e, int node, int mode);
/* Where the stores took effect.  `frame` = frames the engine had been handed (dspfx_frames_submitted) when the store
 * was applied: the store governs every frame from that one on -- always a boundary between two process calls.
 * Copies the applied stores with seq > after_seq, oldest first, into dst[cap]; returns how many (the engine keeps the
 * most recent 4096). */
typedef struct dspfx_param_event {
    uint64_t seq;     /* 1, 2, ... in the order the stores were made on this engine */
    uint64_t frame;
    int32_t node;
    int32_t param;    /* -1: a mode store (value = the mode) */
    float value;
    int32_t reserved;
} dspfx_param_event;
int dspfx_param_log(dspfx_engine *e, dspfx_param_event *dst, int cap, uint64_t after_seq);
/* Frames handed to the process calls since the engine was created (all sub-blocks counted). */
uint64_t dspfx_frames_submitted(const dspfx_engine *e);
/* Reverb::refresh_seconds (reverb.rs:55-71) with D explicit: a NEW zero ring, the same O(1) swap as a slider store (no wait
 * for the device).  A ring longer than the node's capacity has its missing groups allocated inside the call, on the calling
 * thread (DSPFX_ERR_OOM leaves the ring as it was); dspfx_reserve_delay_len beforehand, from any thread, moves that cost off
 * the thread that drives the blocks.  The node's seconds slider (params[1]) is left as it is. */
int dspfx_set_delay_len(dspfx_engine *e, int node, uint32_t delay_len);
/* Capacity hint -- the allocation half of Reverb::refresh_seconds (reverb.rs:60: `circular_buffer::<f32>(num_samples)`) made ahead of
 * the swap half (reverb.rs:70): allocate (not zero, not yet use) the 128-row groups a ring of delay_len samples at `node` would need beyond
 * what the node already has -- e.g. dspfx_delay_len(1.0f, page_round) once after dspfx_chain_set, and no seconds store can
 * allocate again.  Callable from any thread while blocks are running; takes no engine lock, launches nothing. */
int dspfx_reserve_delay_len(dspfx_engine *e, int node, uint32_t delay_len);
/* Give back delay-ring capacity beyond the rings' current lengths (and reservations not yet used) -- what dropping the old ring does in
 * the reference (reverb.rs:70: the previous (source, sink) pair is freed when `*guard` is overwritten), made explicit because here a
 * shorter ring keeps its groups.  Waits for the device. */
int dspfx_ring_trim(dspfx_engine *e);
/* Fir tap reload (fir.rs:153-171).  Like the reference it replaces the taps ONLY: the history is kept (`state`,
 * fir.rs:64-65, is never cleared), and because at most one sample is popped per step (fir.rs:193-197) a history longer
 * than the new tap count STAYS longer -- its oldest samples pair with the taps, i.e. the output is the new convolution
 * delayed by (old length - new length) samples -- while a shorter one goes on filling front-aligned like the warm-up.
 * dspfx_reset (or a new dspfx_chain_set) starts from an empty history. */
int dspfx_set_taps(dspfx_engine *e, int node, const double *taps_reversed, uint32_t n_taps, int mode);
/* How the FIR node's steady-state sweep multiplies (the reference accumulates in f64, fir.rs:201-216; every form below meets
 * the stated 1e-6 relative RMS bar and is bit-exact on data whose products and sums are exact in its operand width):
 *   DSPFX_FIR_PRECISION_HALF     every f32 operand as f16 hi + f16 lo (samples x 2^14, taps x a power of two: 22 significant
 *                                bits), THREE f16 products per term on the matrix pipe, f32 accumulation: 3e-7 relative RMS at
 *                                4096 taps like the others, and the fastest (the sweep becomes HBM-bound).  f16's range is
 *                                narrow: the sweep tracks every channel's peak over the window it swept and lists the tiles with
 *                                a channel at 3.998 or above, or below 2^-13 (-78 dBFS) without being silent; the SPLIT sweep,
 *                                launched right behind it over the listed tiles only (usually none), redoes those.  Whole
 *                                128-frame slices while the tap tables fit the LDS (<= ~5000 taps);
 *   DSPFX_FIR_PRECISION_SPLIT    every f32 operand split exactly into three bf16 parts, six bf16 products per term: any f32
 *                                range (bf16 has f32's exponent), 1.5 x faster than F32, 1.5 x slower than HALF;
 *   DSPFX_FIR_PRECISION_F32      f32 products on the f32 matrix pipe (v_mfma_f32_32x32x2_f32), always;
 *   DSPFX_FIR_PRECISION_DEFAULT  HALF (round 4; DSPFX_FIR_HALF=0 in the environment makes it SPLIT, DSPFX_FIR_SPLIT=0 F32).
 * Takes effect from the next block; history and taps are untouched. */
typedef enum dspfx_fir_precision {
    DSPFX_FIR_PRECISION_DEFAULT = 0,
    DSPFX_FIR_PRECISION_F32 = 1,
    DSPFX_FIR_PRECISION_SPLIT = 2,
    DSPFX_FIR_PRECISION_HALF = 3
} dspfx_fir_precision;
int dspfx_set_fir_precision(dspfx_engine *e, int node, int precision);
/* Zero every node's DSP state (fresh nodes); parameters are kept.  Asynchronous: the clears are queued on the stream
 * the engine was last driven on, behind the blocks in flight there (delay rings are not rewritten at all: their next D
 * frames read zeros, see dspfx_set_param). */
int dspfx_reset(dspfx_engine *e);

/* ---- the hot path ------------------------------------------------------ */
/* One block through the whole chain for all N channels == what N reference
 * graphs do in `Perform::perform` x chain length (node.rs:267-352).
 *   in    : device ptr, [n_frames][N] f32
 *   side  : device ptr or NULL, [n_frames][N]: port "b" of ADD/MIX nodes
 *           (NULL = unconnected port = zeros, node.rs:288)
 *   out   : device ptr, [n_frames][N] (may alias `in`)
 *   mix   : device ptr or NULL, [n_frames] f32: receives sum over this
 *           engine's channels of `out` per frame (the un-normalised mix bus,
 *           node.rs:181-183 before the division; deterministic order)
 *   stream: hipStream_t as void* (NULL = default stream); the call is
 *           asynchronous on that stream.
 * n_frames <= max_frames; DISTORT/Fuzz needs n_frames % 128 == 0 (it is
 * block-global over BUF_SIZE, distort.rs:146-172). */
int dspfx_process(dspfx_engine *e, const float *in, const float *side, float *out, float *mix,
                  uint32_t n_frames, void *stream);
/* dspfx_process with the Output node complete: mix[f] = (sum over this engine's channels of out[f][c]) /
 * dspfx_link_divisor(n_connected) -- nodes/output.rs:215-249 feeding collect_and_average (node.rs:162-194); n_connected
 * = 0 leaves the un-normalised sum (what a rank hands to dspfx_mix_allreduce).  The bus of THIS block, ready when the
 * block's samples are: the chain launch itself finishes the sum in its last workgroups (the workgroup that completes a
 * slice of partial sums reduces it, the one that completes the last slice writes the bus), no further kernel runs.
 * Fixed summation order: bit-identical from run to run and to every other form of the bus in this header. */
int dspfx_process_bus(dspfx_engine *e, const float *in, const float *side, float *out, float *mix, uint32_t n_frames,
                      uint64_t n_connected, void *stream);
/* One connected control port (`as_input` slider, dsp-stuff-derive/src/lib.rs:122-161): `param` is
 * the slider's index in dspfx_node_desc.params (GAIN level 0; DISTORT level 0; OVERDRIVE boost 0,
 * drive 1, level 2; MIX ratio 0; SIGNAL_GEN amplitude 0, frequency 1); `signal` is a device buffer in the sample layout.  Per sample the
 * slider takes lo + (hi-lo)*clamp((x+1)/2, 0, 1) over its reference range; the first value of each
 * 128-frame block is latched per channel and keeps applying once the port is disconnected
 * (lib.rs:148-151) until dspfx_set_param overwrites it.  Every DISTORT mode takes the port, Fuzz included
 * (distort.rs:176-180 maps it before the mode switch; fuzz zips it per sample, 154-160). */
typedef struct dspfx_ctl {
    int32_t node;
    int32_t param;
    const float *signal;
} dspfx_ctl;
int dspfx_process_ctl(dspfx_engine *e, const float *in, const float *side, float *out, float *mix,
                      uint32_t n_frames, const dspfx_ctl *ctl, int n_ctl, void *stream);
/* Placement tuning against the caller's own buffers.  Large delay rings are tables of separately allocated
 * 128-row groups, and how fast a group streams depends on where it landed physically RELATIVE to the sample
 * buffers it is streamed with (placement).  dspfx_chain_set already keeps the fastest of up to 2x
 * candidate groups, judged with scratch buffers; this call repeats that with the real chain kernels reading
 * `in` and writing `out` -- the buffers the host will keep using -- and keeps the fastest groups again.
 * `in` / `out` are laid out like a block of max_frames frames; n_frames <= 128 of it are streamed per probe (`out` is
 * overwritten).  About 3 s at 94 GiB.  DSP state is PRESERVED -- filter state is snapshotted and restored, every ring
 * group's rows are parked while it is probed and end up at the same ring position, the rows the probe blocks overwrite
 * in every OTHER delay ring and in FIR histories are parked and put back -- so a live host can call it again after it
 * re-allocated its buffers (flush the mix pipeline first).  Results never change, only speed. */
int dspfx_tune_placement(dspfx_engine *e, const float *in, const float *side, float *out, uint32_t n_frames,
                         void *stream);
/* Page-locked host memory for the blocks handed to dspfx_process_host.  From ordinary (pageable) buffers the two
 * copies of a block run one after the other; from these buffers dspfx_process_host cuts the block into channel
 * parts and overlaps upload, kernel and download (both directions of the bus busy; DESIGN.md, host buffers).  A host
 * that gathers its pipes into one block anyway should gather into these. */
int dspfx_host_alloc(size_t bytes, void **out);
int dspfx_host_free(void *p);
/* Same with HOST buffers (what a Rust `process(&[f32], &mut [f32])` holds):
 * H2D copy, process, D2H copy, synchronous. */
int dspfx_process_host(dspfx_engine *e, const float *in, const float *side, float *out, float *mix,
                       uint32_t n_frames);
/* Pipelined mix bus.  dspfx_process(mix != NULL) / dspfx_process_bus finish the bus inside the chain launch (a few
 * microseconds at its tail; a chain that ends in a FIR node, an odd block length or DSPFX_MIX_TAIL=0 take two small
 * kernels behind it instead).  The forms below move even that off the block's own launch, at the price of delivering
 * the bus late:
 *   dspfx_process_partials(stream A): the chain, leaving per-wavefront partial sums in one of two
 *       engine-owned buffers (n_frames must not exceed the shortest delay line);
 *   dspfx_mix_collect(stream B): B waits for that chain kernel, reduces the partials into
 *       mix[n_frames] in fixed order.  The next block's chain kernel on A does not wait for it
 *       (it only waits, two blocks later, before reusing the same partial buffer).
 * Every dspfx_process_partials must be followed by exactly one dspfx_mix_collect. */
int dspfx_process_partials(dspfx_engine *e, const float *in, const float *side, float *out,
                           uint32_t n_frames, void *stream);
int dspfx_mix_collect(dspfx_engine *e, float *mix, uint32_t n_frames, void *stream);
/* Mix bus pipelined INSIDE the chain kernel: no second stream, no events, no extra launches.  The launch of
 * block k also runs, in its first 65 workgroups, the slice reduction of block k-1's partials and the final
 * reduction (+ the Output hop when n_connected != 0) of block k-2, so `mix` receives the bus of the block
 * submitted TWO calls earlier (it is not written by the first two calls after a flush / reset; it may be NULL
 * there).  Same reduction tree as dspfx_process(mix) and dspfx_mix_collect: bit-identical sums.  n_frames must
 * stay the same between flushes and must not exceed the shortest delay line.
 * dspfx_mixpipe_flush drains the pipeline with stand-alone kernels: mix_older <- the block before the last one
 * (may be NULL when only one block is in flight), mix_newer <- the last block. */
int dspfx_process_mixpipe(dspfx_engine *e, const float *in, const float *side, float *out, float *mix,
                          uint32_t n_frames, uint64_t n_connected, void *stream);
int dspfx_mixpipe_flush(dspfx_engine *e, float *mix_older, float *mix_newer, uint64_t n_connected, void *stream);
/* Output-node hop of the mix bus (node.rs:189-191): mix[f] /= link_divisor(n_connected),
 * in place on the device; call after the cross-GPU all-reduce with the GLOBAL channel count. */
int dspfx_mix_finish(dspfx_engine *e, float *mix, uint32_t n_frames, uint64_t n_connected, void *stream);

/* ---- the mix bus across GPUs ------------------------------------------------------------------------
 * Channels shard over the GPUs of a node with no data-path exchange; the one collective of the path is the mix bus:
 * the Output node's sum over ALL channels (nodes/output.rs:215-249 feeding node.rs:162-194).  Each rank owns one
 * engine; its un-normalised bus (dspfx_process_bus / dspfx_process(mix) / dspfx_process_mixpipe with n_connected = 0) is summed
 * over the ranks by ONE exchange of n_frames floats, then divided by f32(0.0001 + N_total).
 *
 * One process per GPU (several ranks may also share one device).  Rank 0 calls dspfx_comm_unique_id and hands the
 * DSPFX_COMM_ID_BYTES bytes to the other ranks over whatever channel the host already has (the Rust host's control socket, a
 * file, MPI); every rank then calls dspfx_comm_create(device, n_ranks, rank, id) -- collectively, it blocks until all ranks
 * have joined (DSPFX_COMM_TIMEOUT_MS, default 60 s).  Two backends, chosen by the rank that makes the id (DSPFX_COMM_BACKEND):
 *   mailbox (default)  a one-shot all-reduce by direct peer writes: every rank owns a mailbox in its device memory, opened by
 *                      its peers through hipIpc handles (exchanged via a shared-memory file named by the id: one node, which
 *                      is all xGMI spans); an exchange is ONE kernel of one workgroup per rank that writes its n_frames
 *                      {value, sequence} granules into every peer's mailbox over xGMI, then adds the n_ranks vectors of its
 *                      own mailbox IN RANK ORDER, ((0 + x_0) + x_1) + ..., and applies the Output hop.  The sum is the same
 *                      bits on every rank and from run to run by construction; latency is one peer write + one poll (a few
 *                      microseconds), what the 512-byte, latency-bound exchange wants (a ring or tree only adds hops).  Every
 *                      wait is bounded: a peer that never arrives gives NaNs and an error from the next call, not a hang.
 *                      Up to 16 ranks, n_frames <= 2048.
 *   rccl               ONE ncclAllReduce(sum, float, n_frames) in place + the Output hop.  RCCL is loaded at run time (the copy
 *                      already mapped into the process, else librccl.so.1): a host that never asks for it needs no RCCL.
 *                      Deterministic for a given rank count only as far as RCCL's topology search is.
 * n_ranks = 1 is allowed (id may be NULL: no exchange runs, the calls still divide).
 * dspfx_mix_allreduce is asynchronous on `stream`, in place on `mix` (device, n_frames f32), and applies the Output hop when
 * n_connected != 0: mix[f] = (sum over ranks of mix[f]) / dspfx_link_divisor(n_connected).  Calls on one communicator must be
 * made in the same order by every rank (they are counted). */
typedef struct dspfx_comm dspfx_comm;
#define DSPFX_COMM_ID_BYTES 128
int dspfx_comm_unique_id(void *id_out);
int dspfx_comm_create(int device, int n_ranks, int rank, const void *id, dspfx_comm **out);
void dspfx_comm_destroy(dspfx_comm *c);
int dspfx_comm_size(const dspfx_comm *c);
int dspfx_comm_rank(const dspfx_comm *c);
const char *dspfx_comm_last_error(const dspfx_comm *c);
/* "mailbox", "rccl" or "single" (one rank without an id). */
const char *dspfx_comm_backend(const dspfx_comm *c);
int dspfx_mix_allreduce(dspfx_engine *e, dspfx_comm *c, float *mix, uint32_t n_frames, uint64_t n_connected,
                        void *stream);

/* collect_and_average for a port with n_srcs connected pipes (node.rs:162-194), element-wise on whole
 * blocks: dst = (0 + srcs[0] + srcs[1] + ...) / f32(0.0001 + n_srcs), added in the order given.
 * n_srcs = 0 gives zeros (an unconnected port); n_srcs = 1 is the plain hop.  Device buffers in the
 * engine's sample layout; dst may alias one of the sources.  This is the only piece a graph with
 * fan-in needs besides the chain engines (dsp-stuff_amd/graph.py). */
int dspfx_link_average(dspfx_engine *e, const float *const *srcs, int n_srcs, float *dst, uint32_t n_frames,
                       void *stream);

/* ---- a whole graph in one kernel ------------------------------------------------------------------
 * The reference evaluates a saved graph (DSPConfig) node by node, every link a pipe through memory
 * (node.rs:267-352).  For a DAG of at most DSPFX_GRAPH_MAX_NODES fusable nodes (every kind except FIR and
 * Distort/Fuzz) the engine instead compiles ONE kernel for the graph at run time: node outputs live in
 * registers, every port's collect_and_average (node.rs:162-194) is arithmetic on them, and a block costs one
 * read of the Input node's buffer and one write of the Output node's, whatever the wiring.
 *
 * `nodes` are given in an order in which every link goes forward (src < dst).  A link connects the output of
 * node `src` (or DSPFX_GRAPH_INPUT: the block passed as `in`; DSPFX_GRAPH_INPUT2: the block passed as `side`;
 * DSPFX_GRAPH_ZERO: a connected pipe that carries zeros, the unselected output of a demux) to port `port` of node `dst` (dst == n_nodes: the Output
 * node, whose only port is MAIN; its value is the block written to `out`).  A port with k links averages
 * them in the order given, (0 + x1 + ... + xk) / f32(0.0001 + k); a port without links reads zeros (main,
 * "b") or keeps its slider value (slider ports).  Ports: DSPFX_PORT_MAIN, DSPFX_PORT_SIDE (port "b" of
 * ADD / MIX), DSPFX_PORT_SLIDER + k (the `as_input` port of slider k, dsp-stuff-derive/src/lib.rs:135-153).
 * The engine's link_flags do not apply (every hop is explicit), `side` of the process calls is DSPFX_GRAPH_INPUT2 and
 * control ports cannot be passed to dspfx_process_ctl.  Needs channels % 64 == 0 (whole waves).
 * DSPFX_ERR_UNSUPPORTED: the graph cannot be fused (too many nodes, a FIR / Fuzz node, channel count) or the
 * run-time compiler is unavailable: cut it into a series of such kernels (DSPFX_PORT_RAW, DSPFX_GRAPH_INPUT2: segment_plan in
 * dsp-stuff_amd/graph.py and include/dspfx_graph.hpp) or evaluate it run by run (graph.py).
 * dspfx_chain_set returns the engine to chain mode.  (dspfx_chain_set itself uses the same generated kernel for a
 * run of 9..16 fusable nodes without Add / Mix on engines above 131072 channels: one launch instead of two.) */
#define DSPFX_GRAPH_MAX_NODES 16
#define DSPFX_GRAPH_INPUT (-1)
#define DSPFX_GRAPH_ZERO (-2)
/* a second block from memory: the buffer passed as `side` to the process calls (must then be non-null).  Lets a graph
 * that was cut into consecutive kernels carry a signal AROUND a node that has a kernel of its own -- the dry path
 * beside a FIR cabinet: the kernel after the FIR node reads the FIR output as its Input and the dry signal here. */
#define DSPFX_GRAPH_INPUT2 (-3)
/* Regions of a graph that was cut into several kernels exchange more than two signals: a generated kernel may read up to
 * DSPFX_GRAPH_MAX_IO blocks and write up to DSPFX_GRAPH_MAX_IO blocks.  Inputs: DSPFX_GRAPH_INPUT (block 0),
 * DSPFX_GRAPH_INPUT2 (block 1), DSPFX_GRAPH_INPUT_N(k) for block k (= -(2 + k) from block 2 on); a link into an Add / Mix "b" port or a slider
 * port from one of them is a side input / control signal read from memory.  Outputs: dst == n_nodes is output block 0
 * (the Output node: `out`), dst == n_nodes + m output block m -- averaged like any port, or DSPFX_PORT_RAW to hand one
 * signal over untouched.  Blocks beyond `in` / `side` / `out` are passed with dspfx_process_io. */
#define DSPFX_GRAPH_MAX_IO 16
#define DSPFX_GRAPH_INPUT_N(k) ((k) == 0 ? DSPFX_GRAPH_INPUT : (k) == 1 ? DSPFX_GRAPH_INPUT2 : -(2 + (k)))   /* link source of input block k */
#define DSPFX_PORT_MAIN 0
#define DSPFX_PORT_SIDE 1
#define DSPFX_PORT_SLIDER 2
/* OR into `port`: the port's only link, taken as it is (no averaging, no division).  For cutting a large graph into
 * consecutive kernels at a point where a single signal crosses: the first kernel's Output link is RAW, the next
 * engine reads that buffer as its Input (dsp-stuff_amd/graph.py, segment_plan). */
#define DSPFX_PORT_RAW 256
typedef struct dspfx_graph_link {
    int32_t src;    /* producing node index, DSPFX_GRAPH_INPUT, DSPFX_GRAPH_INPUT2 or DSPFX_GRAPH_ZERO */
    int32_t dst;    /* consuming node index, or n_nodes for the Output node */
    int32_t port;   /* DSPFX_PORT_* of the consumer (| DSPFX_PORT_RAW) */
} dspfx_graph_link;
int dspfx_graph_set(dspfx_engine *e, const dspfx_node_desc *nodes, int n_nodes, const dspfx_graph_link *links,
                    int n_links);
/* dspfx_process for a graph engine with several input / output blocks: ins[k] = input block k (ins[0] = `in`, ins[1] =
 * `side`), outs[m] = output block m (outs[0] = `out`).  Entries the graph does not use may be NULL; n_ins, n_outs <=
 * DSPFX_GRAPH_MAX_IO.  `mix` sums output block 0. */
int dspfx_process_io(dspfx_engine *e, const float *const *ins, int n_ins, float *const *outs, int n_outs, float *mix,
                     uint32_t n_frames, void *stream);
/* The translation unit dspfx_graph_set would compile for this graph (the generated `struct Prog`; it includes
 * csrc/graph_kernel.hip.h), NUL-terminated into dst[cap].  Needs no engine and no device: for inspection and for
 * checking the generator where there is no GPU (without one every division is written in its IEEE form, since the
 * exact-division check runs on the device).  DSPFX_ERR_INVALID: bad graph or cap too small. */
int dspfx_graph_source(const dspfx_node_desc *nodes, int n_nodes, const dspfx_graph_link *links, int n_links,
                       char *dst, size_t cap);

/* ---- DSP state (parity tests; the reference never saves it) --- */
/* Size in bytes of node `node`'s exported state:
 *   BIQUAD 4*N f32 [x1|x2|y1|y2][N]; LOW/HIGH_PASS N f32; REVERB D*N f32
 *   [D][N] oldest sample first; others 0;
 *   FIR: the reference's `state: VecDeque<f64>` (fir.rs:64-65) as it stands -- a 32-byte header {u64 samples pushed
 *   since empty, u64 held = the deque's length, u32 VecDeque capacity, u32 VecDeque head, u32 n_taps, u32 0} followed
 *   by the held samples [held][N] f32, oldest first.  held < n_taps while the deque fills, == n_taps in steady state,
 *   > n_taps after a reload with a shorter impulse response: the size changes with the node's history, so ask right
 *   before exporting.  dspfx_state_import takes such a blob of any length (the size must match the blob's own header). */
int64_t dspfx_state_size(const dspfx_engine *e, int node);
int dspfx_state_export(dspfx_engine *e, int node, void *host_dst, size_t size);
int dspfx_state_import(dspfx_engine *e, int node, const void *host_src, size_t size);

/* ---- utilities --------------------------------------------------------- */
/* Synthetic white noise, identical integer hash on CPU and GPU:
 * dst[f][c] = noise(seed, channel_offset + c, n_abs0 + f), dst device [n_frames][N]. */
int dspfx_fill_noise(dspfx_engine *e, float *dst, uint32_t n_frames, uint32_t n_abs0, uint32_t seed,
                     void *stream);
/* Block until everything queued by this engine on `stream` has finished. */
int dspfx_sync(dspfx_engine *e, void *stream);
/* Human-readable plan of the current chain (stages, kernels, bytes/sample). */
int dspfx_describe(const dspfx_engine *e, char *dst, size_t cap);
/* Division by a wave-uniform constant c (the link divisor, SoftClip's 3.0, a clip level)
 * is evaluated as (float)((double)x * (1.0/c)) when that is bit-identical to IEEE f32
 * x / c for EVERY one of the 2^32 possible x.  That holds for every c that is not an even
 * integer (only those have exact ties among their subnormal quotients: csrc/chain_kernels.hip.h,
 * div_c); the engine takes it for granted there and runs this exhaustive check for even
 * integers.  The call runs the check on the device for ANY c and returns the number of
 * mismatching inputs (0 => the fast form is exact for c), so the rule itself is testable. */
int dspfx_verify_fast_division(int device, float c, uint64_t *mismatches);
/* The Tanh / Sin / Atan modes (distort.rs:109,117,125; overdrive.rs:38; chebyshev.rs:34,40; signal_gen.rs:64)
 * evaluate in f64 and round once.  The engine's own f64 tanh (func 0) / sin (func 1) / atan (func 2) are cheaper than the math library's; this compares the two
 * over all 2^32 inputs on the device: *mismatches = inputs whose f32 results differ, *max_ulp = the largest
 * distance among them (a handful of near-tie inputs, 1 ulp).  func 3: the f64 exp of Fuzz (distort.rs:159).
 * func 4..64: Fuzz's per-lane divisions (distort.rs:158,167,171) as f64 products with an IEEE fallback for
 * subnormal quotients, against IEEE division on 2^32 (numerator, hashed divisor) pairs; must report 0. */
int dspfx_verify_libm(int device, int func, uint64_t *mismatches, uint32_t *max_ulp);
/* Kernel timing for the roofline report: when enabled, every stage's main kernel
 * launch is bracketed by HIP events on the stream it is launched on.  read()
 * synchronises those events and returns, for the stage with the largest total,
 * the summed kernel time and the number of launches (and optionally resets). */
int dspfx_profile_enable(dspfx_engine *e, int enable /* 0 = off; n > 0 = on, pre-creating events for n launches */);
int dspfx_profile_read(dspfx_engine *e, double *total_ms, uint32_t *launches, char *kernel_name, size_t cap,
                       int reset);
/* Algorithmic HBM bytes per channel-sample of the current chain at n_frames. */
double dspfx_algorithmic_bytes_per_sample(const dspfx_engine *e, uint32_t n_frames);

#ifdef __cplusplus
}
#endif
#endif /* DSPFX_H */
