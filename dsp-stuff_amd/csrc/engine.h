// engine.h -- what the translation units of libdspfx.so share: the engine object behind the C ABI (include/dspfx.h) and the
// host-side functions that more than one of them calls.
//   dspfx.hip      lifecycle, parameter stores, run_subblock and the process calls
//   plan.hip       which kernels serve a chain: exact-division decision, variant selection, background specialisation, plan()
//   host_pipe.hip  dspfx_process_host (pinned staging, overlapped upload / kernel / download) and its allocator
//   state_util.hip DSP state export / import, fan-in averaging, utilities (noise, checks, profiling read-out, dspfx_describe)
//   jit.hip        run-time specialisation (hiprtc) of the chain kernels and the generator of whole-graph kernels
//   placement.hip  delay-ring placement tuning (setup-time probe, dspfx_tune_placement)
//   comm.hip       the mix bus across GPUs: RCCL through dlopen, dspfx_comm_*, dspfx_mix_allreduce
// Not installed: the public interface is include/dspfx.h alone.
#pragma once
#include "../../include/dspfx.h"

#include <hip/hip_runtime.h>
#include <hip/hiprtc.h>
#include <dlfcn.h>

#include <algorithm>
#include <cmath>
#include <cstdarg>
#include <cstddef>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <deque>
#include <atomic>
#include <chrono>
#include <map>
#include <mutex>
#include <condition_variable>
#include <deque>
#include <atomic>
#include <thread>
#include <memory>
#include <string>
#include <vector>

#include "aux_kernels.h"
#include "fir_kernels.h"
#include "variants.h"
#include "graph_kernel.hip.h"   // GraphArgs

using namespace dspfx;

namespace dspfx_host {


enum StageType { ST_FUSED = 0, ST_FUZZ = 1, ST_FIR = 2 };

struct Node {
    dspfx_node_desc d{};
    // BIQUAD: normalised coefficients (biquad.rs:62-76)
    float a1 = 0, a2 = 0, b0 = 0, b1 = 0, b2 = 0;
    float *state = nullptr;   // BIQUAD [4][N], LOW/HIGH_PASS [1][N]
    // REVERB: the ring as separately allocated 128-row groups + the device copy of the pointer table
    // `groups` is the ring's CAPACITY: a ring of D rows uses the first ceil(D / 128) of them; groups once allocated stay
    // (a shorter ring keeps them for the next longer one), and entry g of the device table never changes once written,
    // so a length change appends -- in stream order, behind the blocks in flight, which only read the entries of THEIR ring.
    std::vector<float *> groups;
    float **d_groups = nullptr;
    size_t table_cap = 0;           // entries d_groups has room for
    float *probe_group = nullptr;   // placement probes: the node stands in as a one-group ring made of this group
    size_t group_floats = 0;
    int ring_replaced = 0;    // groups re-allocated by the placement probe
    size_t state_bytes = 0;
    uint32_t D = 0, pos = 0;  // REVERB
    // REVERB: frames still to come whose taps read +0.0 whatever the rows hold -- a NEW zero ring of the same length
    // (Reverb::refresh_seconds on a slider change, dspfx_reset, dspfx_set_delay_len with an unchanged D) is this counter set
    // to D: O(1), no memset of up to 94 GiB, placement kept.  Counted down by every block (SlotArgs::zero_rows).
    uint32_t zero_left = 0;
    // REVERB: the seconds slider is known -- given with the node (params[1] > 0) or stored since.  A STORE of exactly 0.0 is a value
    // like any other (reverb.rs:58: max(0, 128) = a 128-sample ring), while 0 in a descriptor means "not given: keep delay_len".
    bool seconds_given = false;
    // FIR
    std::vector<double> taps;      // reversed, as given
    FirState fir;
    // control ports: per-channel latched slider values (derive lib.rs:148)
    float *latch[3] = {nullptr, nullptr, nullptr};
    int latch_valid = 0;
    const float *ctl_now[3] = {nullptr, nullptr, nullptr};   // signals of the call being launched
};

// A small engine's chain shape being specialised in the background (jit.hip: async_jit_submit).  The engine runs the
// interpreter until `ready`, then adopts the kernels at its next block boundary.  Shared between the engine (which may drop
// it: abandoned) and the compiler thread (which owns nothing of the engine).
struct JitKernel;
struct AsyncJit {
    std::atomic<int> ready{0};           // 1: compiled (whatever could be), -1: no run-time compiler / headers
    std::atomic<bool> abandoned{false};  // the engine re-planned or is gone: skip if not started yet
    std::atomic<int> state{0};           // 0 queued, 1 being compiled, 2 finished or skipped
    int device = 0, n_slots = 0;
    int sigs[dspfx::MAX_SLOTS];
    // which kernels of the shape: the standard one <f_std, cpl_std>, the one with control ports <f_mod, cpl_std, mod>, the
    // time-sliced one <32, cpl_ts>, the guarded time-sliced one for the channels a whole-wave launch leaves over
    bool want_std = false, want_mod = false, want_ts = false, want_tail = false;
    int f_std = 16, f_mod = 8, cpl_std = 1, cpl_ts = 1;
    const JitKernel *k_std = nullptr, *k_mod = nullptr, *k_ts = nullptr, *k_tail = nullptr;
    std::string headers_dir;             // where the kernel headers are, as the submitting thread saw it
    std::string cache_dir;               // ... and the on-disk cache of code objects ("": none)
};

// DSPFX_VARIANT="f=8,cpl=2,static=1" narrows the choice (tuning / A-B runs).
struct Pref {
    int f = -1, cpl = -1, stat = -1;
};

// Every DSPFX_* switch that bears on planning or on the per-block path, read from the environment ONCE per setup call
// (dspfx_engine_create, dspfx_chain_set / dspfx_graph_set) and kept on the engine: nothing a process call, a slider
// store or the background compiler's hand-over does calls getenv -- a host may be changing its environment at that
// moment, and the calls are not free (tests/test_abi_cpu.py greps for it).  -1 = not set.
struct EnvSwitches {
    int xcd_remap = -1;          // DSPFX_XCD_REMAP
    int mix_tail = -1;           // DSPFX_MIX_TAIL (0: stand-alone bus kernels)
    int fast_div = -1;           // DSPFX_FAST_DIV (0: IEEE division everywhere)
    int jit = -1, jit_async = -1;   // DSPFX_JIT, DSPFX_JIT_ASYNC
    int ts_tail = -1;            // DSPFX_TS_TAIL
    int menu_ring_reserve = -1;  // DSPFX_MENU_RING_RESERVE (0: never reserve a menu-fresh delay node's slider ring at chain set; 1: whenever it fits; unset: when cheap)
    bool has_variant = false;    // DSPFX_VARIANT set at all
    int variant_ts = -1;         // its ts= field
    bool variant_static0 = false;   // it contains "static=0"
    Pref pref;                   // its f= / cpl= / static= fields
    std::string headers_dir;     // DSPFX_KERNEL_HEADERS ("": the embedded text)
    std::string cache_dir;       // the on-disk cache of code objects as resolved from DSPFX_DISK_CACHE / DSPFX_CACHE_DIR / XDG_CACHE_HOME / HOME ("": none)
};
EnvSwitches read_env_switches();

struct Stage {
    StageType type;
    int first, count;
    mutable const Variant *var = nullptr;   // ST_FUSED
    mutable const Variant *var_ts = nullptr;   // ST_FUSED, few channels: the time-sliced kernel, used for blocks of exactly 4 * ts frames
    mutable const Variant *var_ts_tail = nullptr;   // ST_FUSED, N % (64 cpl) != 0: the guarded time-sliced kernel for the channels left over (same blocks)
    mutable std::shared_ptr<AsyncJit> async;    // ST_FUSED on the interpreter: its specialisation is on the way (adopted at a block boundary)
    mutable bool jit_failed = false;            // ... and could not be had (no compiler / headers): the best interpreter instantiation serves
    mutable const Variant *var_mod = nullptr;   // ST_FUSED, control ports connected: specialised kernel, asked for on first use
    mutable bool var_mod_tried = false;
    mutable std::shared_ptr<AsyncJit> async_mod;   // ... being compiled in the background
    mutable int mod_two = -1;                   // ... 1 / 0: the control-port interpreter's channels per lane (2 / 1) fixed for the life of this plan when such a job was submitted
    bool fast_div = false;          // all constant divisors of the stage verified (see divisor_is_fast)
};

}  // namespace dspfx_host
using namespace dspfx_host;

struct dspfx_engine {
    dspfx_engine_desc desc{};
    int device = 0;
    EnvSwitches env;                          // the environment as of the last setup call (engine.h: EnvSwitches)
    std::vector<Node> nodes;
    std::vector<Stage> stages;
    bool graph_mode = false;                  // dspfx_graph_set: the nodes form a DAG evaluated by one generated kernel
    bool no_long = false;                     // chain mode: do not fuse more than MAX_SLOTS nodes into one (graph) kernel
    std::vector<dspfx_graph_link> wiring;     // its links, in the caller's order
    std::string err;
    float hop_div = 1.0f;
    float *mixpart = nullptr;
    float *mixpart_b = nullptr;   // second-stage scratch [128][max_frames] (one per stream of use: inline / deferred)
    float *mixpart_b2 = nullptr;
    unsigned *mt_tickets = nullptr;   // same-block bus inside the chain launch: [MIX_SLICES + 1] arrival counters, zero between launches
    size_t mixpart_cols = 0;
    // pipelined mix bus (dspfx_process_partials / dspfx_mix_collect): double-buffered partials
    float *mixpart2[2] = {nullptr, nullptr};
    hipEvent_t ev_chain[2] = {nullptr, nullptr}, ev_red[2] = {nullptr, nullptr};
    bool red_pending[2] = {false, false}, collect_due = false;
    int flip = 0;
    uint32_t part_stride[2] = {0, 0}, part_frames[2] = {0, 0};
    float *partials_override = nullptr;   // set while a deferred-mix block is being launched
    // in-kernel pipelined mix bus (dspfx_process_mixpipe): block k's launch also runs stage 2 of block k-1 and
    // stage 3 of block k-2.  mp_count = blocks submitted since the last flush.
    uint64_t mp_count = 0;
    uint32_t mp_frames = 0, mp_rows[2] = {0, 0};
    float *mp_mix_now = nullptr;          // where the launch being built delivers block k-2's bus
    float mp_div_now = 0.0f;
    bool mp_building = false;
    float bus_div_now = 0.0f;             // dspfx_process_bus: the Output hop's divisor for the block being launched (0: none)
    // channel window of the current run_subblock call (pipelined host path): channels [win_c0, win_c0 + win_n), 0 = all;
    // win_last marks the call that finishes the block (ring positions advance once)
    uint32_t win_c0 = 0, win_n = 0;
    bool win_last = true;
    hipStream_t hs_in = nullptr, hs_out = nullptr, hs_run = nullptr;
    std::vector<hipEvent_t> hev;
    uint32_t ctl_tile_frames = 0;         // dspfx_process_ctl: frames of the caller's whole block (tile stride)
    // dspfx_process_io: input blocks 2.. and output blocks 1.. of the call being launched (graph engines), and the float
    // offset of the sub-block being launched
    const float *io_in[GRAPH_IO] = {};
    float *io_out[GRAPH_IO] = {};
    size_t io_off = 0;
    // staging for dspfx_process_host
    float *h_in = nullptr, *h_side = nullptr, *h_out = nullptr, *h_mix = nullptr;
    const Variant *tail = nullptr, *dyn = nullptr, *tail_mod = nullptr, *dyn_mod = nullptr, *dyn_mod2 = nullptr;
    bool has_fuzz = false;
    uint32_t min_delay = 0xffffffffu;
    bool has_siggen = false;   // a SIGNAL_GEN wraps its clock per 128-frame block: sub-launches start on block boundaries
    mutable bool jit_unavailable = false;   // a run-time specialised kernel was wanted and could not be had (headers / hiprtc missing)
    uint64_t div_n = 0;   // cached Output-hop divisor (dspfx_mix_finish)
    float div_v = 0.0f;
    // profiling: event pairs per stage
    bool profiling = false;
    std::vector<std::vector<std::pair<hipEvent_t, hipEvent_t>>> prof;   // [stage][launch]
    std::vector<hipEvent_t> ev_pool;
    // ---- threads and streams (include/dspfx.h, "Threads and streams") ----------------------------------------------
    // api_mu serialises every entry point that touches the engine; a slider store from another thread never waits
    // for it: it is queued under pend_mu and applied by whoever holds api_mu next, at a block boundary.
    mutable std::recursive_mutex api_mu;
    int api_depth = 0;                        // nesting of entry points (dspfx_process_ctl -> dspfx_process): only the outermost drains
    mutable std::mutex pend_mu;               // pending, pub_kinds, log, next_seq
    struct Store { uint64_t seq; int node, param; float value; };   // param -1: a mode store (value = the mode)
    std::deque<Store> pending;
    std::vector<int> pub_kinds;               // node kinds as of the last chain / graph set: validation without api_mu
    std::deque<dspfx_param_event> log;        // the stores already applied, oldest first (bounded)
    uint64_t next_seq = 1;
    uint64_t frames_submitted = 0;            // frames handed to the process calls so far
    // the stream the DSP state was last touched on: every state write (biquad reset, dspfx_reset, ...) is queued on
    // it, and a call on a different stream first waits for an event recorded there
    hipStream_t cur_stream = nullptr;
    bool cur_stream_set = true;               // the null stream to begin with: setup-time writes go there
    hipEvent_t ev_order = nullptr;
    // ---- delay-ring capacity (dspfx.hip: ring_reserve / ring_resize) ------------------------------------------------
    // A slider store that lengthens a ring needs groups the ring does not have yet.  The thread that MAKES the store (the
    // reference's GUI thread, which allocates the new ring itself: reverb.rs:55-71) allocates them into ring_pool[node]
    // before it queues the store; the thread that drives the blocks only takes them over at the block boundary.
    // alloc_mu serialises allocating threads (held across hipMalloc calls), pool_mu guards the containers (never held
    // across a HIP call), pub_rev (under pend_mu) is what a storing thread may know about a REVERB node without api_mu.
    std::mutex alloc_mu, pool_mu;
    std::vector<std::vector<float *>> ring_pool;      // [node]: allocated groups not yet part of the node's ring
    std::vector<void *> retired;                      // group tables replaced while blocks were in flight: freed when the device is idle
    struct PubReverb { float seconds = 0.0f; int mode = 0; uint32_t D = 0; size_t have = 0; bool given = false; };
    std::vector<PubReverb> pub_rev;                   // [node] (REVERB nodes only carry meaning)
    uint64_t chain_gen = 0;                           // bumped by every chain / graph set (under pend_mu)
    mutable std::mutex err_mu;                        // err (also kept per calling thread: dspfx_last_error)
};

namespace dspfx_host {

struct JitKernel {
    Variant var;            // launch == nullptr: launched through `fn`
    hipModule_t module = nullptr;
    hipFunction_t fn = nullptr;
    std::string name;
    int vgprs = 0;          // registers per lane the compiler allocated (occupancy: 512 / vgprs waves per SIMD)
};

// One link of the program being generated; node indices are local to the stage.  raw: the only link into its port and
// taken as it is -- a hop of a chain engine whose DSPFX_LINK_* flag is off.
struct GLink {
    int src, dst, port;
    bool raw;
};

// JIT_MIN_CHANNELS: from here on DSPFX_JIT_ASYNC=0 / DSPFX_VARIANT mean "compile inside dspfx_chain_set" (below: "interpreter only");
// by default every engine gets its kernels from the background compiler (jit.hip: jit_policy).  TS_MAX_CHANNELS: up to here a
// whole 128-frame block of a chain of up to three nodes MAY go through the time-sliced kernel (pick_ts_variant: 65536 for longer
// chains, and for short ones where the standard kernel takes two channels per lane) -- and up to here the bus' first stage
// leaves one row per wave, whatever kernel runs (dspfx.hip: rows_per_wave).
constexpr uint32_t JIT_MIN_CHANNELS = 16384, TS_MAX_CHANNELS = 98304;
// from here on the specialised standard kernels take two channels per lane in the tiled layout (below: one, except short chains
// between 65536 and 131072 channels: jit_std_cpl; plan.hip has the sweep)
constexpr uint32_t STATIC_CPL2_MIN_CHANNELS = 229376;

// ---- errors
int fail(dspfx_engine *e, int code, const char *fmt, ...);
#define HIPCHK(e, call)                                                                         \
    do {                                                                                        \
        hipError_t err__ = (call);                                                              \
        if (err__ != hipSuccess)                                                                \
            return fail(e, err__ == hipErrorOutOfMemory ? DSPFX_ERR_OOM : DSPFX_ERR_HIP,        \
                        "%s failed: %s", #call, hipGetErrorString(err__));                      \
    } while (0)

// ---- dspfx.hip
inline hipEvent_t take_event(dspfx_engine *e) {
    if (!e->ev_pool.empty()) {
        hipEvent_t ev = e->ev_pool.back();
        e->ev_pool.pop_back();
        return ev;
    }
    hipEvent_t ev = nullptr;
    (void)hipEventCreate(&ev);
    return ev;
}

struct ProfScope {   // brackets one kernel launch with events on its own stream
    dspfx_engine *e;
    hipStream_t s;
    hipEvent_t a = nullptr, b = nullptr;
    size_t stage;
    ProfScope(dspfx_engine *e_, size_t stage_, hipStream_t s_) : e(e_), s(s_), stage(stage_) {
        if (!e->profiling) return;
        a = take_event(e);
        b = take_event(e);
        (void)hipEventRecord(a, s);
    }
    ~ProfScope() {
        if (!a) return;
        (void)hipEventRecord(b, s);
        if (e->prof.size() <= stage) e->prof.resize(stage + 1);
        e->prof[stage].emplace_back(a, b);
    }
};

bool divisor_is_fast(float c, bool have_device = true, bool forced_off = false);
bool node_divisors_fast(const Node &n, bool have_device = true, bool forced_off = false);
bool stage_fast_div(const dspfx_engine *e, const Stage &st, bool have_device = true);
bool fusable(const Node &n);
int node_hop(const dspfx_engine *e, int idx);
inline const Pref &read_pref(const dspfx_engine *e) { return e->env.pref; }
void async_jit_submit(const std::shared_ptr<AsyncJit> &job);   // jit.hip: background specialisation for small engines
bool async_jit_wait(const std::shared_ptr<AsyncJit> &job, int wait_ms);   // ... until the compiler is done with `job` (bounded)
int validate_node(dspfx_engine *e, const dspfx_node_desc &d);
int plan(dspfx_engine *e);
void collect_variants(std::vector<const Variant *> &out);
void adopt_async_jit(dspfx_engine *e, const Stage &st);      // plan.hip: run_subblock calls it at a block boundary
void request_mod_kernel(dspfx_engine *e, const Stage &st);   // plan.hip: the stage's control-port kernel, on first use
int ring_rows_copy(dspfx_engine *e, Node &n, uint32_t r0, uint32_t nrows, char *host, bool to_host);
inline size_t ring_groups_for(uint32_t D) { return ((size_t)D + RING_GROUP_ROWS - 1) / RING_GROUP_ROWS; }
int ring_reserve(dspfx_engine *e, int node, uint64_t gen, size_t want_groups, size_t keep_free, bool quiet, size_t max_share);   // any thread, no api_mu
int ring_resize(dspfx_engine *e, int node, uint32_t D, hipStream_t s);          // api_mu held
void recompute_min_delay(dspfx_engine *e);
int run_subblock(dspfx_engine *e, const float *in, const float *side, float *out, float *mix, uint32_t nframes, uint32_t tile_frames,
                 hipStream_t stream);
int bind_stream(dspfx_engine *e, hipStream_t s);
int quiesce(dspfx_engine *e);
int settle_null_stream(dspfx_engine *e);
int drain_pending(dspfx_engine *e, hipStream_t s);
// Every entry point that touches an engine: take api_mu; the outermost one applies the queued stores first.  With a
// stream (the process calls, tuning) the state is bound to it before anything is queued.
struct ApiScope {
    dspfx_engine *e;
    int rc = DSPFX_OK;
    bool outer = false;
    ApiScope(dspfx_engine *e_, bool has_stream = false, hipStream_t s = nullptr) : e(e_) {
        e->api_mu.lock();
        outer = e->api_depth++ == 0;
        if (hipSetDevice(e->device) != hipSuccess) rc = fail(e, DSPFX_ERR_HIP, "hipSetDevice(%d) failed", e->device);
        if (has_stream && rc == DSPFX_OK) rc = bind_stream(e, s);
        if (outer && rc == DSPFX_OK) rc = drain_pending(e, e->cur_stream_set ? e->cur_stream : nullptr);
    }
    ~ApiScope() {
        --e->api_depth;
        e->api_mu.unlock();
    }
};

// ---- jit.hip
enum JitMode { JIT_MEMORY = 0, JIT_DISK = 1, JIT_COMPILE = 2 };      // how far jit_get goes: this process' table, + the disk cache, + the compiler
enum JitPolicy { JP_OFF = 0, JP_SYNC = 1, JP_ASYNC = 2 };
JitPolicy jit_policy(const dspfx_engine *e);
const JitKernel *jit_get(int device, const int (&sigs)[MAX_SLOTS], int n_slots, int f, int cpl, bool mod, bool ts, bool guard, int mode);
void stage_sigs(const dspfx_engine *e, const Stage &st, int (&sigs)[MAX_SLOTS]);
int jit_std_cpl(const dspfx_engine *e, int n_slots);
int jit_std_f(const dspfx_engine *e, bool mod, int n_slots);
std::string jit_cache_dir();
std::string jit_headers_dir();
// While alive, this thread's run-time compiler look-ups take their header / cache directories from the engine's snapshot
// instead of the environment (the background thread is handed them with its job in the same way).
struct JitDirScope {
    const std::string *old_hdr, *old_cache;
    explicit JitDirScope(const dspfx_engine *e);
    ~JitDirScope();
};
void jit_arm_exit_guard();   // the calling thread waits for a background compile in flight when it ends (jit.hip: ExitGuard)
extern std::atomic<uint64_t> g_jit_compiled, g_jit_from_disk, g_jit_disk_written;
std::string jit_cache_rejected();     // "" or the cache directory this process refused to use (not this user's / not private)
int launch_variant(const Variant *v, const ChainArgs &a, unsigned grid, unsigned block, unsigned lds_bytes, hipStream_t s);
const Variant *jit_variant(const dspfx_engine *e, const Stage &st, bool mod, int mode);
const Variant *graph_variant(const dspfx_engine *e, const Stage &st);
int kind_sliders(const dspfx_node_desc &d, float (&lo)[3], float (&hi)[3]);
int graph_input_block(int src);
int validate_graph(dspfx_engine *e, const dspfx_node_desc *nodes, int n_nodes, const dspfx_graph_link *links, int n_links);

// ---- placement.hip
hipError_t big_alloc(void **p, size_t bytes);
int tune_ring(dspfx_engine *e, Node &n);

}  // namespace dspfx_host
