// dspfx.hpp -- C++17 host-side mirror of the reference's operator interface over the C ABI (dspfx.h).
//
// The reference is Rust; its toolchain is absent from the build image, so the host layer above the
// C ABI is written in C++ (and mirrored in Python for the tests).  Names, slider fields, ranges and
// defaults follow dsp-stuff/src/nodes/*.rs; `GpuChain::process` has the shape of
// `SimpleNode::process` (dsp-stuff/src/node.rs:135-146): borrowed input slice(s) in, output slice out,
// node-owned parameters and state.  Errors that the reference turns into panics
// (node.rs:173,271,280) become dspfx::Error exceptions here; nothing falls back to the CPU.
#pragma once
#include <cstddef>
#include <array>
#include <cstdint>
#include <cstring>
#include <stdexcept>
#include <string>
#include <utility>
#include <vector>

#include "dspfx.h"

namespace dspfx {

struct Error : std::runtime_error {
    int status;
    Error(int st, const std::string &msg) : std::runtime_error("dspfx error " + std::to_string(st) + ": " + msg), status(st) {}
};

constexpr std::size_t BUF_SIZE = DSPFX_BUF_SIZE;   // node.rs:257

// nodes/distort.rs:18-28
enum class Mode : int { HardClip = 0, SoftClip, Tanh, RecipSoftClip, Fuzz, Sin, Atan, Square, Chebyshev4 };
// nodes/fir.rs Mode
enum class FirMode : int { Balanced = 0, Average = 1 };

// One node of a chain: the reference node's saved fields.
struct Node {
    dspfx_node_desc d{};
    std::vector<double> taps;   // Fir: time-reversed, as fir.rs:163,168 stores them
};

inline Node make(int kind) {
    Node n;
    if (dspfx_node_defaults(kind, &n.d) != DSPFX_OK) throw Error(DSPFX_ERR_INVALID, "unknown node kind");
    return n;
}
// nodes/gain.rs: slider level 0..=10, default 1.0
inline Node Gain(float level = 1.0f) { Node n = make(DSPFX_GAIN); n.d.params[0] = level; return n; }
// nodes/biquad.rs:18-41: raw sliders -10..=10, normalised by a0 on the engine (biquad.rs:62-76)
inline Node BiQuad(float a0 = 1.0f, float a1 = -0.24f, float a2 = 0.0f, float b0 = 0.758f, float b1 = 0.0f, float b2 = 0.0f) {
    Node n = make(DSPFX_BIQUAD);
    const float p[6] = {a0, a1, a2, b0, b1, b2};
    for (int i = 0; i < 6; ++i) n.d.params[i] = p[i];
    return n;
}
inline Node LowPass(float ratio = 0.5f) { Node n = make(DSPFX_LOW_PASS); n.d.params[0] = ratio; return n; }
inline Node HighPass(float ratio = 0.5f) { Node n = make(DSPFX_HIGH_PASS); n.d.params[0] = ratio; return n; }
// nodes/reverb.rs: a RESTORED node -- refresh_seconds has run (dsp-stuff-derive/src/lib.rs:319-337), the ring has
// reverb.rs:58's length for `seconds`; the seconds slider travels with the node (params[1]) so that a later slider store --
// which swaps in a new zero ring, decay included (reverb.rs:19, 55-71) -- refreshes to the same length
inline Node Reverb(float seconds = 0.5f, float decay = 0.5f, bool page_round = false) {
    Node n = make(DSPFX_REVERB);
    n.d.params[0] = decay;
    n.d.params[1] = seconds;
    n.d.mode = page_round ? 1 : 0;
    n.d.delay_len = dspfx_delay_len(seconds, page_round ? 1 : 0);
    return n;
}
// a node fresh from the menu: make_buffer()'s ring under the 0.5 s slider (reverb.rs:44-52) = dspfx_node_defaults: 128 samples,
// or 1024 under the page-rounded reading of rivulet (make_buffer() is refresh_seconds' three calls with 128 for num_samples)
inline Node ReverbFresh(bool page_round = false) {
    Node n = make(DSPFX_REVERB);
    n.d.mode = page_round ? 1 : 0;
    n.d.delay_len = dspfx_delay_len(0.0f, page_round ? 1 : 0);
    return n;
}
// an explicit ring and no seconds slider: a slider store swaps in a zero ring of the same length
inline Node ReverbSamples(std::uint32_t delay_len, float decay = 0.5f) {
    Node n = make(DSPFX_REVERB);
    n.d.params[0] = decay;
    n.d.params[1] = 0.0f;
    n.d.delay_len = delay_len;
    return n;
}
// nodes/distort.rs: level 0..=30 default 0 (bypass), mode default SoftClip
inline Node Distort(float level = 0.0f, Mode mode = Mode::SoftClip) {
    Node n = make(DSPFX_DISTORT);
    n.d.params[0] = level;
    n.d.mode = static_cast<int>(mode);
    return n;
}
inline Node Overdrive(float boost = 0.0f, float drive = 0.0f, float level = 0.0f) {
    Node n = make(DSPFX_OVERDRIVE);
    n.d.params[0] = boost; n.d.params[1] = drive; n.d.params[2] = level;
    return n;
}
inline Node Chebyshev(float level_pos = 0.0f, float level_neg = 0.0f) {
    Node n = make(DSPFX_CHEBYSHEV);
    n.d.params[0] = level_pos; n.d.params[1] = level_neg;
    return n;
}
// nodes/fir.rs: impulse response h[0..T) in natural order; stored reversed like fir.rs:163,168
inline Node Fir(const std::vector<double> &impulse_response, FirMode mode = FirMode::Balanced) {
    Node n = make(DSPFX_FIR);
    n.taps.assign(impulse_response.rbegin(), impulse_response.rend());
    n.d.mode = static_cast<int>(mode);
    return n;
}
inline Node Add() { return make(DSPFX_ADD); }
inline Node Mix(float ratio = 0.5f) { Node n = make(DSPFX_MIX); n.d.params[0] = ratio; return n; }
// nodes/envelope.rs:27-30: peak envelope follower, attack / release in frames
inline Node Envelope(float attack = 0.0f, float release = 0.0f) {
    Node n = make(DSPFX_ENVELOPE);
    n.d.params[0] = attack; n.d.params[1] = release;
    return n;
}
// nodes/signal_gen.rs:41-55: a source (no "in" port) -- as a chain node it replaces the signal
enum class SignalMode : int { Sine = DSPFX_SIG_SINE, Triangle = DSPFX_SIG_TRIANGLE, Square = DSPFX_SIG_SQUARE, Constant = DSPFX_SIG_CONSTANT };
inline Node SignalGen(float amplitude = 0.5f, float frequency = 100.0f, SignalMode mode = SignalMode::Sine) {
    Node n = make(DSPFX_SIGNAL_GEN);
    n.d.params[0] = amplitude; n.d.params[1] = frequency;
    n.d.mode = static_cast<int>(mode);
    return n;
}

// The translation unit Engine::set_graph would compile for this graph (dspfx.h: dspfx_graph_source); needs no device.
inline std::string graph_source(const std::vector<Node> &nodes, const std::vector<dspfx_graph_link> &links) {
    std::vector<dspfx_node_desc> d;
    for (const Node &n : nodes) d.push_back(n.d);
    std::string out(std::size_t{1} << 18, '\0');
    const int rc = dspfx_graph_source(d.data(), static_cast<int>(d.size()), links.data(), static_cast<int>(links.size()),
                                      out.data(), out.size());
    if (rc != DSPFX_OK) throw Error(rc, dspfx_strerror(rc));
    out.resize(std::strlen(out.c_str()));
    return out;
}

// N independent mono channels through one chain.
class Engine {
  public:
    Engine(std::uint32_t channels, std::uint32_t max_frames = DSPFX_BUF_SIZE,
           std::uint32_t link_flags = DSPFX_LINK_INTERNAL | DSPFX_LINK_INPUT, int device = 0,
           std::uint32_t tile_channels = 0, std::uint64_t channel_offset = 0)
        : channels_(channels) {
        dspfx_engine_desc d{DSPFX_ABI_VERSION, device, channels, max_frames, link_flags, tile_channels, channel_offset};
        const int rc = dspfx_engine_create(&d, &e_);
        if (rc != DSPFX_OK) throw Error(rc, dspfx_strerror(rc));
    }
    ~Engine() { dspfx_engine_destroy(e_); }
    Engine(const Engine &) = delete;
    Engine &operator=(const Engine &) = delete;
    Engine(Engine &&o) noexcept : e_(std::exchange(o.e_, nullptr)), channels_(o.channels_) {}

    void set_chain(const std::vector<Node> &nodes) {
        std::vector<dspfx_node_desc> d;
        for (const Node &n : nodes) {
            d.push_back(n.d);
            if (n.d.kind == DSPFX_FIR) {
                d.back().taps = n.taps.data();
                d.back().n_taps = static_cast<std::uint32_t>(n.taps.size());
            }
        }
        chk(dspfx_chain_set(e_, d.data(), static_cast<int>(d.size())));
    }
    // specialised kernels are compiled in the background and adopted at a block boundary: wait for them (dspfx.h)
    bool kernels_ready(int wait_ms = 60000) {
        const int rc = dspfx_kernels_ready(e_, wait_ms);
        if (rc < 0) chk(rc);
        return rc == 1;
    }
    // a whole DAG as one generated kernel (dspfx.h, dspfx_graph_set); Error::status == DSPFX_ERR_UNSUPPORTED when it cannot be fused
    void set_graph(const std::vector<Node> &nodes, const std::vector<dspfx_graph_link> &links) {
        std::vector<dspfx_node_desc> d;
        for (const Node &n : nodes) d.push_back(n.d);
        chk(dspfx_graph_set(e_, d.data(), static_cast<int>(d.size()), links.data(), static_cast<int>(links.size())));
    }
    // Slider / mode stores are safe from another thread while this one is inside process(): queued, applied at the next
    // block boundary in order (dspfx.h, "threads").  set_param_seq returns the store's sequence number, param_log where
    // the stores took effect.
    void set_param(int node, int param, float v) { chk(dspfx_set_param(e_, node, param, v)); }
    std::uint64_t set_param_seq(int node, int param, float v) {
        std::uint64_t seq = 0;
        chk(dspfx_set_param_seq(e_, node, param, v, &seq));
        return seq;
    }
    void set_mode(int node, Mode m) { chk(dspfx_set_mode(e_, node, static_cast<int>(m))); }
    std::vector<dspfx_param_event> param_log(std::uint64_t after_seq = 0) {
        std::vector<dspfx_param_event> ev(4096);
        const int n = dspfx_param_log(e_, ev.data(), static_cast<int>(ev.size()), after_seq);
        if (n < 0) chk(n);
        ev.resize(static_cast<std::size_t>(n));
        return ev;
    }
    std::uint64_t frames_submitted() const { return dspfx_frames_submitted(e_); }
    void set_delay_len(int node, std::uint32_t d) { chk(dspfx_set_delay_len(e_, node, d)); }
    void reserve_delay_len(int node, std::uint32_t d) { chk(dspfx_reserve_delay_len(e_, node, d)); }   // capacity hint (any thread)
    void ring_trim() { chk(dspfx_ring_trim(e_)); }
    /// DSPFX_FIR_PRECISION_DEFAULT / _F32 / _SPLIT / _HALF: how a FIR node's steady-state sweep multiplies (dspfx.h)
    void set_fir_precision(int node, dspfx_fir_precision p) { chk(dspfx_set_fir_precision(e_, node, static_cast<int>(p))); }
    void reset() { chk(dspfx_reset(e_)); }
    // device buffers, asynchronous on `stream`
    void process(const float *in, float *out, std::uint32_t n_frames, const float *side = nullptr,
                 float *mix = nullptr, void *stream = nullptr) {
        chk(dspfx_process(e_, in, side, out, mix, n_frames, stream));
    }
    // the Output node in the same launch: mix = this block's bus / link_divisor(n_connected) (0: the un-normalised sum)
    void process_bus(const float *in, float *out, float *mix, std::uint32_t n_frames, std::uint64_t n_connected,
                     const float *side = nullptr, void *stream = nullptr) {
        chk(dspfx_process_bus(e_, in, side, out, mix, n_frames, n_connected, stream));
    }
    // host buffers ([n_frames][channels]), synchronous
    void process_host(const float *in, float *out, std::uint32_t n_frames, const float *side = nullptr,
                      float *mix = nullptr) {
        chk(dspfx_process_host(e_, in, side, out, mix, n_frames));
    }
    void mix_finish(float *mix, std::uint32_t n_frames, std::uint64_t n_connected, void *stream = nullptr) {
        chk(dspfx_mix_finish(e_, mix, n_frames, n_connected, stream));
    }
    // the mix bus across GPUs: sum this rank's un-normalised bus over the communicator's ranks (ONE RCCL all-reduce of
    // n_frames floats, in place, asynchronous on `stream`), then the Output hop with the GLOBAL channel count
    void mix_allreduce(dspfx_comm *comm, float *mix, std::uint32_t n_frames, std::uint64_t n_connected, void *stream = nullptr) {
        chk(dspfx_mix_allreduce(e_, comm, mix, n_frames, n_connected, stream));
    }
    // re-tune the delay rings' placement against the buffers the host will keep using (DSP state is kept)
    void tune_placement(const float *in, float *out, std::uint32_t n_frames, const float *side = nullptr, void *stream = nullptr) {
        chk(dspfx_tune_placement(e_, in, side, out, n_frames, stream));
    }
    // collect_and_average over several pipes (node.rs:162-194): dst = (0 + srcs...) / f32(0.0001 + n)
    void link_average(const std::vector<const float *> &srcs, float *dst, std::uint32_t n_frames, void *stream = nullptr) {
        chk(dspfx_link_average(e_, srcs.data(), static_cast<int>(srcs.size()), dst, n_frames, stream));
    }
    std::uint32_t channels() const { return channels_; }
    dspfx_engine *raw() { return e_; }

  private:
    void chk(int rc) {
        if (rc != DSPFX_OK) throw Error(rc, dspfx_last_error(e_));
    }
    dspfx_engine *e_ = nullptr;
    std::uint32_t channels_;
};

// The mix bus' communicator: one per process / GPU (include/dspfx.h).  Rank 0 calls Comm::unique_id() and hands the
// bytes to every rank over the host's own channel; every rank then constructs its Comm (collective).
class Comm {
  public:
    using Id = std::array<unsigned char, DSPFX_COMM_ID_BYTES>;
    static Id unique_id() {
        Id id{};
        const int rc = dspfx_comm_unique_id(id.data());
        if (rc != DSPFX_OK) throw Error(rc, dspfx_comm_last_error(nullptr));
        return id;
    }
    Comm(int device, int n_ranks, int rank, const Id *id = nullptr) {
        const int rc = dspfx_comm_create(device, n_ranks, rank, id ? id->data() : nullptr, &c_);
        if (rc != DSPFX_OK) throw Error(rc, dspfx_comm_last_error(nullptr));
    }
    ~Comm() { dspfx_comm_destroy(c_); }
    Comm(const Comm &) = delete;
    Comm &operator=(const Comm &) = delete;
    int size() const { return dspfx_comm_size(c_); }
    int rank() const { return dspfx_comm_rank(c_); }
    dspfx_comm *raw() { return c_; }

  private:
    dspfx_comm *c_ = nullptr;
};

// Reference-shaped node: what a `GpuChain: SimpleNode` in the Rust host does per block.
// process(input, output) takes one 128-frame block per channel bank laid out [frame][channel]
// (channels == 1 reproduces the reference's mono node exactly); the host's Perform wrapper has
// already averaged the input pipes (node.rs:290-299), so only the hops BETWEEN the fused nodes are
// applied here (DSPFX_LINK_INTERNAL).
class GpuChain {
  public:
    GpuChain(std::vector<Node> chain, std::uint32_t channels = 1, int device = 0)
        : eng_(channels, DSPFX_BUF_SIZE, DSPFX_LINK_INTERNAL, device) {
        eng_.set_chain(chain);
    }
    static const char *title() { return "GPU chain"; }
    static const char *cfg_name() { return "gpu_chain"; }
    void process(const float *input, float *output, std::size_t n_frames = BUF_SIZE) {
        eng_.process_host(input, output, static_cast<std::uint32_t>(n_frames));
    }
    Engine &engine() { return eng_; }

  private:
    Engine eng_;
};

}  // namespace dspfx
