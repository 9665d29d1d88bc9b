// C++ parity test of the host-side mirror (include/dspfx.hpp) against the CPU oracle (oracle/): reads like
// a test the reference could have had for its nodes.  Exit code 0 = pass.  Built and run by
// tests/test_cpp_host.py (g++ only: links libdspfx.so and liboracle.so).
#include <cmath>
#include <cstdio>
#include <atomic>
#include <chrono>
#include <cstring>
#include <thread>
#include <vector>
#include <csignal>
#include <execinfo.h>
#include <unistd.h>

#include "../../include/dspfx.hpp"
#include "../../include/dspfx_ir.hpp"
#include "../../oracle/dspfx_oracle.h"

static int ulp(float a, float b) {
    if (a != a && b != b) return 0;
    int32_t ia, ib;
    std::memcpy(&ia, &a, 4);
    std::memcpy(&ib, &b, 4);
    if (ia < 0) ia = -(ia & 0x7fffffff);
    if (ib < 0) ib = -(ib & 0x7fffffff);
    return std::abs(ia - ib);
}

// a crash anywhere (this program, the library, its background compiler thread) leaves its call stack on stderr for the test's report
static void on_crash(int sig) {
    void *frames[64];
    const int n = backtrace(frames, 64);
    const char msg[] = "FAIL: fatal signal; call stack:\n";
    (void)!write(2, msg, sizeof msg - 1);
    backtrace_symbols_fd(frames, n, 2);
    signal(sig, SIG_DFL);
    raise(sig);
}

int main(int argc, char **argv) {
    signal(SIGSEGV, on_crash);
    signal(SIGBUS, on_crash);
    signal(SIGABRT, on_crash);
    setvbuf(stdout, nullptr, _IOLBF, 0);        // what was printed before a crash is in the report
    const bool expect_no_device = argc > 1 && std::strcmp(argv[1], "--expect-no-device") == 0;
    if (argc > 2 && std::strcmp(argv[1], "--ir") == 0) {
        // f3 from C++: test_host --ir file.wav [blocks] -- the impulse response through include/dspfx_ir.hpp (fir.rs:86-173:
        // decode, channel average, resample to 48 kHz), Fir(...) of include/dspfx.hpp (reversal, fir.rs:163,168), 64 channels of
        // hashed noise through the engine; prints the taps' count and every output sample of channels 0 and 63 as hex floats
        try {
            using namespace dspfx;
            const std::vector<double> h = load_impulse_response(argv[2]);
            const uint32_t N = 64, blocks = argc > 3 ? (uint32_t)std::atoi(argv[3]) : 6;
            Engine eng(N, BUF_SIZE, 0);
            eng.set_chain({Fir(h)});
            std::vector<float> x(BUF_SIZE * N), y(BUF_SIZE * N);
            std::printf("taps %zu\n", h.size());
            for (uint32_t blk = 0; blk < blocks; ++blk) {
                for (uint32_t f = 0; f < BUF_SIZE; ++f)
                    for (uint32_t c = 0; c < N; ++c) x[f * N + c] = orc_noise(0x5EED0004u, c, blk * BUF_SIZE + f);
                eng.process_host(x.data(), y.data(), BUF_SIZE);
                for (uint32_t f = 0; f < BUF_SIZE; ++f) std::printf("%a %a\n", (double)y[f * N + 0], (double)y[f * N + 63]);
            }
            return 0;
        } catch (const std::exception &e) {
            std::printf("FAIL: %s\n", e.what());
            return 1;
        }
    }
    try {
        using namespace dspfx;
        const uint32_t N = 200, blocks = 6;
        std::vector<Node> chain = {Gain(0.8f), BiQuad(1.0f, -1.8f, 0.81f, 0.0025f, 0.005f, 0.0025f),
                                   Distort(3.0f, Mode::SoftClip), ReverbSamples(256, 0.5f), HighPass(0.25f)};
        Engine eng(N, BUF_SIZE, DSPFX_LINK_INTERNAL | DSPFX_LINK_INPUT);
        if (expect_no_device) { std::printf("FAIL: engine created without a device\n"); return 1; }
        eng.set_chain(chain);
        // oracle twins, one per channel
        std::vector<std::vector<orc_node *>> twins(N);
        for (uint32_t c = 0; c < N; ++c) {
            orc_node *g = orc_node_new(ORC_GAIN); orc_node_set_param(g, 0, 0.8f);
            orc_node *b = orc_node_new(ORC_BIQUAD);
            const float p[6] = {1.0f, -1.8f, 0.81f, 0.0025f, 0.005f, 0.0025f};
            for (int i = 0; i < 6; ++i) orc_node_set_param(b, i, p[i]);
            orc_node *d = orc_node_new(ORC_DISTORT); orc_node_set_param(d, 0, 3.0f); orc_node_set_mode(d, ORC_DIST_SOFT_CLIP);
            orc_node *r = orc_node_new(ORC_REVERB); orc_node_set_param(r, 0, 0.5f); orc_reverb_set_len(r, 256);
            orc_node *h = orc_node_new(ORC_HIGH_PASS); orc_node_set_param(h, 0, 0.25f);
            twins[c] = {g, b, d, r, h};
        }
        std::vector<float> x(BUF_SIZE * N), y(BUF_SIZE * N), mix(BUF_SIZE), xc(BUF_SIZE), yc(BUF_SIZE);
        int worst = 0;
        double mix_err = 0;
        for (uint32_t blk = 0; blk < blocks; ++blk) {
            for (uint32_t f = 0; f < BUF_SIZE; ++f)
                for (uint32_t c = 0; c < N; ++c) x[f * N + c] = orc_noise(0x5EED0001u, c, blk * BUF_SIZE + f);
            eng.process_host(x.data(), y.data(), BUF_SIZE, nullptr, mix.data());
            std::vector<double> msum(BUF_SIZE, 0.0);
            for (uint32_t c = 0; c < N; ++c) {
                for (uint32_t f = 0; f < BUF_SIZE; ++f) xc[f] = x[f * N + c];
                orc_chain_run(twins[c].data(), 5, 3, xc.data(), nullptr, yc.data(), BUF_SIZE, BUF_SIZE);
                for (uint32_t f = 0; f < BUF_SIZE; ++f) {
                    worst = std::max(worst, ulp(y[f * N + c], yc[f]));
                    msum[f] += yc[f];
                }
            }
            for (uint32_t f = 0; f < BUF_SIZE; ++f) mix_err = std::max(mix_err, std::fabs(msum[f] - mix[f]));
        }
        // the reference-shaped single-channel node
        GpuChain node({Gain(2.0f)}, 1);
        float in1[BUF_SIZE], out1[BUF_SIZE];
        for (uint32_t f = 0; f < BUF_SIZE; ++f) in1[f] = orc_noise(7, 0, f);
        node.process(in1, out1);
        for (uint32_t f = 0; f < BUF_SIZE; ++f)
            if (out1[f] != in1[f] * 2.0f) { std::printf("FAIL: GpuChain gain\n"); return 1; }
        // a chain handed over as a graph (every link explicit, the Output node's hop included) is the chain engine's
        // block divided once more by f32(0.0001 + 1): same bits, from C++
        const uint32_t NG = 256;
        std::vector<Node> short_chain = {Gain(0.8f), BiQuad(1.0f, -1.8f, 0.81f, 0.0025f, 0.005f, 0.0025f), HighPass(0.25f)};
        Engine as_chain(NG, BUF_SIZE, DSPFX_LINK_INTERNAL | DSPFX_LINK_INPUT), as_graph(NG, BUF_SIZE, 0);
        as_chain.set_chain(short_chain);
        as_graph.set_graph(short_chain, {{DSPFX_GRAPH_INPUT, 0, DSPFX_PORT_MAIN}, {0, 1, DSPFX_PORT_MAIN}, {1, 2, DSPFX_PORT_MAIN},
                                         {2, 3, DSPFX_PORT_MAIN}});
        std::vector<float> xg(BUF_SIZE * NG), yc2(BUF_SIZE * NG), yg(BUF_SIZE * NG);
        for (uint32_t f = 0; f < BUF_SIZE; ++f)
            for (uint32_t c = 0; c < NG; ++c) xg[f * NG + c] = orc_noise(0x5EED0009u, c, f);
        as_chain.process_host(xg.data(), yc2.data(), BUF_SIZE);
        as_graph.process_host(xg.data(), yg.data(), BUF_SIZE);
        const float hop = dspfx_link_divisor(1);
        for (size_t i = 0; i < yg.size(); ++i)
            if (ulp(yg[i], (0.0f + yc2[i]) / hop) != 0) { std::printf("FAIL: graph-as-chain at %zu\n", i); return 1; }
        if (graph_source(short_chain, {{DSPFX_GRAPH_INPUT, 0, DSPFX_PORT_MAIN}, {0, 3, DSPFX_PORT_MAIN}}).find("struct Prog") == std::string::npos) {
            std::printf("FAIL: graph_source\n");
            return 1;
        }
        // the mix bus across GPUs through the C ABI: a ONE-rank communicator made from a real unique id (so RCCL's
        // ncclCommInitRank / ncclAllReduce really run on this GPU), in place on a page-locked buffer, then the Output hop
        for (const char *backend : {"mailbox", "rccl"}) {
            setenv("DSPFX_COMM_BACKEND", backend, 1);            // read by the rank that makes the id
            Comm::Id id = Comm::unique_id();
            unsetenv("DSPFX_COMM_BACKEND");
            Comm comm(0, 1, 0, &id);
            if (comm.size() != 1 || comm.rank() != 0) { std::printf("FAIL: comm size/rank\n"); return 1; }
            if (std::strcmp(dspfx_comm_backend(comm.raw()), backend) != 0) { std::printf("FAIL: backend %s\n", dspfx_comm_backend(comm.raw())); return 1; }
            void *pinned = nullptr;
            if (dspfx_host_alloc(BUF_SIZE * sizeof(float), &pinned) != DSPFX_OK) { std::printf("FAIL: host_alloc\n"); return 1; }
            float *pm = static_cast<float *>(pinned);
            for (uint32_t f = 0; f < BUF_SIZE; ++f) pm[f] = mix[f];
            const std::uint64_t n_total = 3 * N;
            eng.mix_allreduce(comm.raw(), pm, BUF_SIZE, n_total);
            dspfx_sync(eng.raw(), nullptr);
            const float div = dspfx_link_divisor(n_total);
            for (uint32_t f = 0; f < BUF_SIZE; ++f)
                if (ulp(pm[f], mix[f] / div) != 0) { std::printf("FAIL: mix_allreduce at %u: %g vs %g\n", f, pm[f], mix[f] / div); return 1; }
            Comm solo(0, 1, 0);                                  // no id: no RCCL communicator, the call still divides
            for (uint32_t f = 0; f < BUF_SIZE; ++f) pm[f] = mix[f];
            eng.mix_allreduce(solo.raw(), pm, BUF_SIZE, n_total);
            dspfx_sync(eng.raw(), nullptr);
            for (uint32_t f = 0; f < BUF_SIZE; ++f)
                if (ulp(pm[f], mix[f] / div) != 0) { std::printf("FAIL: solo mix_allreduce at %u\n", f); return 1; }
            dspfx_host_free(pinned);
            std::printf("mix_allreduce over a 1-rank %s communicator ok\n", backend);
        }
        // GpuBank (host/rust/src/gpu_bank.rs) -- the N-channel engine as a node of the reference graph -- makes exactly these C
        // calls in this order; here they run from C++ against the oracle: N input ports gathered into one page-locked [128][N]
        // block, ONE dspfx_process_host, N output ports scattered, the `mix` port = sum / f32(0.0001 + N) (node.rs:189-191).
        {
            const uint32_t NB = 64, nblk = 4;
            dspfx_engine *be = nullptr;
            const dspfx_engine_desc bd{DSPFX_ABI_VERSION, 0, NB, BUF_SIZE, DSPFX_LINK_INTERNAL, 0, 0};
            if (dspfx_engine_create(&bd, &be) != DSPFX_OK) { std::printf("FAIL: bank engine\n"); return 1; }
            const std::vector<Node> bchain = {BiQuad(1.0f, -1.8f, 0.81f, 0.0025f, 0.005f, 0.0025f), Distort(3.0f, Mode::SoftClip), Reverb(0.5f, 0.5f),
                                              BiQuad(1.0f, -1.98f, 0.9801f, 0.99f, -1.98f, 0.99f), Gain(0.5f)};
            std::vector<dspfx_node_desc> bdesc;
            for (const Node &n : bchain) bdesc.push_back(n.d);
            if (dspfx_chain_set(be, bdesc.data(), (int)bdesc.size()) != DSPFX_OK) { std::printf("FAIL: bank chain: %s\n", dspfx_last_error(be)); return 1; }
            void *pg = nullptr, *ps = nullptr;
            if (dspfx_host_alloc((size_t)BUF_SIZE * NB * sizeof(float), &pg) != DSPFX_OK || dspfx_host_alloc((size_t)BUF_SIZE * NB * sizeof(float), &ps) != DSPFX_OK) { std::printf("FAIL: bank host_alloc\n"); return 1; }
            const float bdiv = dspfx_link_divisor(NB);
            float *gather = static_cast<float *>(pg), *scatter = static_cast<float *>(ps);
            std::vector<std::vector<orc_node *>> bt(NB);
            for (uint32_t c = 0; c < NB; ++c) {
                orc_node *b0 = orc_node_new(ORC_BIQUAD), *b1 = orc_node_new(ORC_BIQUAD);
                const float p0[6] = {1.0f, -1.8f, 0.81f, 0.0025f, 0.005f, 0.0025f}, p1[6] = {1.0f, -1.98f, 0.9801f, 0.99f, -1.98f, 0.99f};
                for (int i = 0; i < 6; ++i) { orc_node_set_param(b0, i, p0[i]); orc_node_set_param(b1, i, p1[i]); }
                orc_node *ds = orc_node_new(ORC_DISTORT); orc_node_init_param(ds, 0, 3.0f); orc_node_set_mode(ds, ORC_DIST_SOFT_CLIP);
                orc_node *rv = orc_node_new(ORC_REVERB); orc_node_init_param(rv, 0, 0.5f); orc_node_init_param(rv, 1, 0.5f);
                orc_node_after_settings_change(rv);                                   // a restored reverb: 24000 samples
                orc_node *gn = orc_node_new(ORC_GAIN); orc_node_init_param(gn, 0, 0.5f);
                bt[c] = {b0, ds, rv, b1, gn};
            }
            int bworst = 0;
            double bmix_err = 0.0;
            std::vector<float> port_in(BUF_SIZE), port_out(BUF_SIZE), bmix(BUF_SIZE), want(BUF_SIZE);
            for (uint32_t blk = 0; blk < nblk; ++blk) {
                for (uint32_t c = 0; c < NB; ++c)                                     // gather: port in{c} -> column c
                    for (uint32_t f = 0; f < BUF_SIZE; ++f) gather[f * NB + c] = orc_noise(0x5EED0011u, c, blk * BUF_SIZE + f);
                if (dspfx_process_host(be, gather, nullptr, scatter, bmix.data(), BUF_SIZE) != DSPFX_OK) { std::printf("FAIL: bank process: %s\n", dspfx_last_error(be)); return 1; }
                std::vector<double> msum(BUF_SIZE, 0.0);
                for (uint32_t c = 0; c < NB; ++c) {
                    for (uint32_t f = 0; f < BUF_SIZE; ++f) port_in[f] = gather[f * NB + c];
                    orc_chain_run(bt[c].data(), 5, 1, port_in.data(), nullptr, want.data(), BUF_SIZE, BUF_SIZE);   // hops BETWEEN the nodes only
                    for (uint32_t f = 0; f < BUF_SIZE; ++f) {
                        port_out[f] = scatter[f * NB + c];                            // scatter: column c -> port out{c}
                        bworst = std::max(bworst, ulp(port_out[f], want[f]));
                        msum[f] += want[f];
                    }
                }
                for (uint32_t f = 0; f < BUF_SIZE; ++f) {
                    const float mix_port = bmix[f] / bdiv;                            // the `mix` port
                    bmix_err = std::max(bmix_err, std::fabs((double)mix_port - msum[f] / (double)bdiv));
                }
            }
            dspfx_host_free(pg);
            dspfx_host_free(ps);
            dspfx_engine_destroy(be);
            for (auto &v : bt) for (orc_node *n : v) orc_node_free(n);
            if (bworst > 1 || bmix_err > 1e-5) { std::printf("FAIL: GpuBank sequence: max ulp %d, mix err %g\n", bworst, bmix_err); return 1; }
            std::printf("GpuBank call sequence: %u ports, max ulp %d, mix err %.3g\n", NB, bworst, bmix_err);
        }
        // Two host threads, as in the reference: the audio task drives process (here the synchronous host form, block after
        // block) while the GUI thread stores a slider (dsp-stuff-derive/src/lib.rs:487-492).  No lock on the caller's side: the
        // store is queued by the library and lands on a block boundary -- every block of the output was made with ONE level,
        // the levels appear in the order they were stored, and the log says from which frame on each one held.
        {
            const uint32_t NT = 64, nblk = 60;
            Engine te(NT, BUF_SIZE, 0);
            te.set_chain({Gain(1.0f)});
            std::vector<float> tx(BUF_SIZE * NT), ty(BUF_SIZE * NT * nblk);
            for (size_t i = 0; i < tx.size(); ++i) tx[i] = 1.0f + orc_noise(3, (uint32_t)(i % NT), (uint32_t)(i / NT));
            std::atomic<bool> go{false}, done{false};
            std::vector<std::uint64_t> seqs;
            std::thread gui([&] {
                while (!go.load()) std::this_thread::yield();
                for (int k = 1; k <= 25 && !done.load(); ++k) {
                    seqs.push_back(te.set_param_seq(0, 0, 0.25f * (float)k));
                    std::this_thread::sleep_for(std::chrono::microseconds(300));
                }
            });
            std::atomic<int> reads{0};
            std::thread monitor([&] {                            // a third thread only looks: plan text, counters, state sizes
                char text[4096];
                while (!done.load()) {
                    if (dspfx_describe(te.raw(), text, sizeof text) != DSPFX_OK || dspfx_chain_len(te.raw()) != 1 ||
                        dspfx_state_size(te.raw(), 0) != 0) { reads.store(-1000000); return; }
                    (void)te.frames_submitted();
                    reads.fetch_add(1);
                }
            });
            for (uint32_t b = 0; b < nblk; ++b) {
                if (b == 3) go.store(true);
                te.process_host(tx.data(), ty.data() + (size_t)b * BUF_SIZE * NT, BUF_SIZE);
            }
            done.store(true);
            gui.join();
            monitor.join();
            // (on a starved host the monitor may not have got a turn at the engine's lock within the 60 blocks: not a failure)
            if (reads.load() < 0) { std::printf("FAIL: the monitoring thread saw an inconsistent engine\n"); return 1; }
            te.process_host(tx.data(), ty.data(), 0);                    // an entry point: whatever is still queued is applied
            const std::vector<dspfx_param_event> log = te.param_log();
            if (log.size() != seqs.size()) { std::printf("FAIL: %zu stores made, %zu logged\n", seqs.size(), log.size()); return 1; }
            float level = 1.0f;
            size_t li = 0;
            for (uint32_t b = 0; b < nblk; ++b) {
                while (li < log.size() && log[li].frame <= (std::uint64_t)b * BUF_SIZE) {
                    if (log[li].seq != seqs[li] || log[li].frame % BUF_SIZE) { std::printf("FAIL: log entry %zu\n", li); return 1; }
                    level = log[li++].value;
                }
                for (size_t i = 0; i < (size_t)BUF_SIZE * NT; ++i)
                    if (ty[(size_t)b * BUF_SIZE * NT + i] != tx[i] * level) { std::printf("FAIL: block %u is not level %g throughout\n", b, level); return 1; }
            }
            std::printf("slider stores from a second thread: %zu applied on block boundaries, in order\n", log.size());
        }
        // error behaviour: exceptions, not aborts
        bool threw = false;
        try { eng.set_chain({ReverbSamples(64)}); } catch (const Error &e) { threw = e.status == DSPFX_ERR_INVALID; }
        std::printf("max ulp %d, mix err %.3g, invalid chain threw %d\n", worst, mix_err, (int)threw);
        return (worst <= 1 && mix_err < 1e-3 && threw) ? 0 : 1;
    } catch (const dspfx::Error &e) {
        if (expect_no_device && e.status == DSPFX_ERR_NO_DEVICE) { std::printf("ok: %s\n", e.what()); return 0; }
        std::printf("FAIL: %s\n", e.what());
        return 1;
    }
}
