// The C++ saved-graph importer (include/dspfx_graph.hpp).  Built and run by tests/test_cpp_graph.py (g++ only).
//   test_graph <doc.json> --plan                      print the plan handed to dspfx_graph_set (no device needed)
//   test_graph <doc.json> <x.f32> <y.f32> <channels> <frames> [--fir-tolerance]
//        run the graph on the GPU (64 copies of the given channels, process_host, 128-frame blocks; one engine or the
//        series of engines segment_plan makes) and compare with y: prints "max ulp N ..."; exit code 0 when N <= 1
//        (--fir-tolerance: relative error <= 1e-5 of the peak instead, the FIR path's bar)
#include <cmath>
#include <cstdio>
#include <cstring>
#include <fstream>
#include <sstream>
#include <vector>

#include "../../include/dspfx_graph.hpp"

static std::string slurp(const char *path) {
    std::ifstream f(path, std::ios::binary);
    std::stringstream ss;
    ss << f.rdbuf();
    return ss.str();
}
static std::vector<float> floats(const char *path) {
    const std::string s = slurp(path);
    std::vector<float> v(s.size() / 4);
    std::memcpy(v.data(), s.data(), v.size() * 4);
    return v;
}
static int ulp(float a, float b) {
    if (a != a && b != b) return 0;
    int32_t ia, ib;
    std::memcpy(&ia, &a, 4);
    std::memcpy(&ib, &b, 4);
    if (ia < 0) ia = -(ia & 0x7fffffff);
    if (ib < 0) ib = -(ib & 0x7fffffff);
    return std::abs(ia - ib);
}

int main(int argc, char **argv) {
    if (argc >= 2 && std::strncmp(argv[1], "--dump-chain", 12) == 0) {   // a chain written the way File > Save would
        using namespace dspfx;
        const std::vector<Node> chain = {BiQuad(1.0f, -1.8f, 0.81f, 0.0025f, 0.005f, 0.0025f), LowPass(0.3f), Distort(3.0f, Mode::Tanh),
                                         ReverbSamples(24000, 0.4f), Mix(0.25f), Overdrive(2.0f, 0.5f, 0.75f), Fir({0.5, 0.25, -0.125}, FirMode::Average),
                                         Envelope(4.0f, 100.0f), Gain(0.1f)};
        std::printf("%s\n", dump_dspconfig(chain, std::strcmp(argv[1], "--dump-chain-faithful") == 0).c_str());
        return 0;
    }
    if (argc < 3) return 2;
    try {
        using namespace dspfx;
        const SavedGraph g(slurp(argv[1]));
        std::vector<Node> specs;
        std::vector<dspfx_graph_link> links;
        const bool one = g.fused_plan(specs, links);
        std::vector<SavedGraph::Step> steps;
        const bool series = !one && g.segment_plan(steps);
        if (std::strcmp(argv[2], "--plan-regions") == 0) {                 // the general plan (SavedGraph::region_plan)
            std::vector<SavedGraph::RegionStep> rs;
            if (!g.region_plan(rs)) { std::printf("no region plan\n"); return 0; }
            for (const SavedGraph::RegionStep &st : rs) {
                std::printf("step %s %d\n", st.kind == SavedGraph::RegionStep::Region ? "region" : st.kind == SavedGraph::RegionStep::NodeStep ? "node" : "output",
                            st.kind == SavedGraph::RegionStep::Region ? st.n_out : 0);
                for (const SavedGraph::Ref &r : st.in_refs) std::printf("in %d %d\n", r.step, r.block);
                for (const SavedGraph::Ref &r : st.main_refs) std::printf("main %d %d\n", r.step, r.block);
                for (const auto &kv : st.ctl_refs) std::printf("ctl %d %d %d\n", kv.first, kv.second.step, kv.second.block);
                for (const Node &n : st.specs) std::printf("node %d %d %a %u %zu\n", n.d.kind, n.d.mode, n.d.params[0], n.d.delay_len, n.taps.size());
                for (const dspfx_graph_link &l : st.links) std::printf("link %d %d %d\n", l.src, l.dst, l.port);
            }
            return 0;
        }
        if (argc >= 7 && std::strcmp(argv[6], "--regions") == 0) {
            // Run the graph through the region plan from C++: every block lives in page-locked host memory (device-visible,
            // so the g++-only test needs no HIP headers) and the regions exchange blocks through dspfx_process_io.
            std::vector<SavedGraph::RegionStep> rs;
            if (!g.region_plan(rs)) { std::printf("no region plan\n"); return 2; }
            const std::vector<float> x = floats(argv[2]), y = floats(argv[3]);
            const uint32_t C = (uint32_t)std::atoi(argv[4]), frames = (uint32_t)std::atoi(argv[5]), N = 64;
            auto pinned = [&]() {
                void *p = nullptr;
                if (dspfx_host_alloc((size_t)BUF_SIZE * N * sizeof(float), &p) != DSPFX_OK) throw Error(DSPFX_ERR_OOM, "dspfx_host_alloc");
                std::memset(p, 0, (size_t)BUF_SIZE * N * sizeof(float));
                return static_cast<float *>(p);
            };
            float *xin = pinned(), *zeros = pinned(), *final_out = pinned();
            std::vector<Engine> engines;
            std::vector<std::vector<float *>> outs(rs.size());
            std::vector<float *> scratch(rs.size(), nullptr);
            std::vector<int> eng_of(rs.size(), -1);
            for (std::size_t k = 0; k < rs.size(); ++k) {
                const SavedGraph::RegionStep &st = rs[k];
                if (st.kind == SavedGraph::RegionStep::OutputAvg) { outs[k] = {final_out}; continue; }
                if (st.kind == SavedGraph::RegionStep::Region) {
                    engines.emplace_back(N, BUF_SIZE, 0);
                    engines.back().set_graph(st.specs, st.links);
                    for (int m = 0; m < st.n_out; ++m) outs[k].push_back(pinned());
                } else {
                    engines.emplace_back(N, BUF_SIZE, DSPFX_LINK_INTERNAL | (st.main_refs.size() == 1 ? DSPFX_LINK_INPUT : 0u));
                    engines.back().set_chain(st.specs);
                    outs[k].push_back(pinned());
                    if (st.main_refs.size() > 1) scratch[k] = pinned();
                }
                eng_of[k] = (int)engines.size() - 1;
            }
            auto at = [&](const SavedGraph::Ref &r) -> float * { return r.step == -1 ? xin : r.step == -2 ? zeros : outs[(std::size_t)r.step][(std::size_t)r.block]; };
            int worst = 0;
            double max_abs = 0, max_err = 0;
            for (uint32_t f0 = 0; f0 + BUF_SIZE <= frames; f0 += BUF_SIZE) {
                for (uint32_t f = 0; f < BUF_SIZE; ++f)
                    for (uint32_t c = 0; c < N; ++c) xin[f * N + c] = x[(f0 + f) * C + c % C];
                for (std::size_t k = 0; k < rs.size(); ++k) {
                    const SavedGraph::RegionStep &st = rs[k];
                    if (st.kind == SavedGraph::RegionStep::OutputAvg) {
                        std::vector<const float *> srcs;
                        for (const SavedGraph::Ref &r : st.main_refs) srcs.push_back(at(r));
                        engines[0].link_average(srcs, final_out, BUF_SIZE);
                        continue;
                    }
                    Engine &e = engines[(std::size_t)eng_of[k]];
                    if (st.kind == SavedGraph::RegionStep::Region) {
                        std::vector<const float *> ins;
                        for (const SavedGraph::Ref &r : st.in_refs) ins.push_back(at(r));
                        int rc = dspfx_process_io(e.raw(), ins.data(), (int)ins.size(), outs[k].data(), (int)outs[k].size(), nullptr, BUF_SIZE, nullptr);
                        if (rc != DSPFX_OK) throw Error(rc, dspfx_last_error(e.raw()));
                    } else {
                        const float *src = zeros;
                        if (st.main_refs.size() == 1) src = at(st.main_refs[0]);
                        else if (st.main_refs.size() > 1) {
                            std::vector<const float *> srcs;
                            for (const SavedGraph::Ref &r : st.main_refs) srcs.push_back(at(r));
                            e.link_average(srcs, scratch[k], BUF_SIZE);
                            src = scratch[k];
                        }
                        std::vector<dspfx_ctl> ctl;
                        for (const auto &kv : st.ctl_refs) ctl.push_back({0, kv.first, at(kv.second)});
                        int rc = ctl.empty() ? dspfx_process(e.raw(), src, nullptr, outs[k][0], nullptr, BUF_SIZE, nullptr)
                                             : dspfx_process_ctl(e.raw(), src, nullptr, outs[k][0], nullptr, BUF_SIZE, ctl.data(), (int)ctl.size(), nullptr);
                        if (rc != DSPFX_OK) throw Error(rc, dspfx_last_error(e.raw()));
                    }
                }
                dspfx_sync(engines[0].raw(), nullptr);
                const float *yb = outs.back()[0];
                for (uint32_t f = 0; f < BUF_SIZE; ++f)
                    for (uint32_t c = 0; c < N; ++c) {
                        const float want = y[(f0 + f) * C + c % C];
                        worst = std::max(worst, ulp(yb[f * N + c], want));
                        max_abs = std::max(max_abs, (double)std::fabs(want));
                        max_err = std::max(max_err, (double)std::fabs(yb[f * N + c] - want));
                    }
            }
            std::printf("max ulp %d rel err %.3g steps %zu engines %zu (region plan)\n", worst, max_err / max_abs, rs.size(), engines.size());
            const bool loose = argc > 7 && std::strcmp(argv[7], "--fir-tolerance") == 0;
            return (loose ? max_err <= 1e-5 * max_abs : worst <= 1) ? 0 : 1;
        }
        if (std::strcmp(argv[2], "--plan") == 0) {
            if (!one && !series) { std::printf("run by run\n"); return 0; }
            if (!one) {
                for (const SavedGraph::Step &st : steps) {
                    std::printf("step %s %d %d\n", st.kind == SavedGraph::Step::GraphKernel ? "graph" : st.kind == SavedGraph::Step::NodeHop ? "node_hop" : "node",
                                st.in_ref, st.in2_ref);
                    for (const Node &n : st.specs) std::printf("node %d %d %a %u %zu\n", n.d.kind, n.d.mode, n.d.params[0], n.d.delay_len, n.taps.size());
                    for (const dspfx_graph_link &l : st.links) std::printf("link %d %d %d\n", l.src, l.dst, l.port);
                }
                return 0;
            }
            for (const Node &n : specs)
                std::printf("node %d %d %a %a %a %a %a %a %u\n", n.d.kind, n.d.mode, n.d.params[0], n.d.params[1], n.d.params[2],
                            n.d.params[3], n.d.params[4], n.d.params[5], n.d.delay_len);
            for (const dspfx_graph_link &l : links) std::printf("link %d %d %d\n", l.src, l.dst, l.port);
            return 0;
        }
        if (argc < 6 || (!one && !series)) return 2;
        const std::vector<float> x = floats(argv[2]), y = floats(argv[3]);
        const uint32_t C = (uint32_t)std::atoi(argv[4]), frames = (uint32_t)std::atoi(argv[5]), N = 64;
        // one engine, or the series segment_plan made: every step's block goes through host buffers here (a host that
        // keeps its blocks on the device chains dspfx_process calls instead, as dsp-stuff_amd/graph.py does)
        std::vector<Engine> engines;
        std::vector<bool> reads2, is_node;   // (kept for the printout below)
        if (one) {
            engines.emplace_back(N, BUF_SIZE, 0);
            g.install(engines.back());
            reads2.push_back(false);
            is_node.push_back(false);
        } else {
            for (const SavedGraph::Step &st : steps) {
                engines.emplace_back(N, BUF_SIZE, st.kind == SavedGraph::Step::NodeHop ? DSPFX_LINK_INPUT : 0);
                if (st.kind == SavedGraph::Step::GraphKernel) engines.back().set_graph(st.specs, st.links);
                else engines.back().set_chain(st.specs);
                reads2.push_back(st.reads_second_block());
                is_node.push_back(st.kind != SavedGraph::Step::GraphKernel);
            }
        }
        std::vector<int> in_ref, in2_ref;
        if (one) { in_ref = {-1}; in2_ref = {SavedGraph::Step::NO_REF}; }
        for (const SavedGraph::Step &st : steps) { in_ref.push_back(st.in_ref); in2_ref.push_back(st.in2_ref); }
        std::vector<std::vector<float>> bufs(engines.size() + 1, std::vector<float>(BUF_SIZE * N));   // [0] = the Input block, [k+1] = step k's output
        int worst = 0;
        double max_abs = 0, max_err = 0;
        for (uint32_t f0 = 0; f0 + BUF_SIZE <= frames; f0 += BUF_SIZE) {
            for (uint32_t f = 0; f < BUF_SIZE; ++f)
                for (uint32_t c = 0; c < N; ++c) bufs[0][f * N + c] = x[(f0 + f) * C + c % C];
            for (std::size_t k = 0; k < engines.size(); ++k)
                engines[k].process_host(bufs[(std::size_t)(in_ref[k] + 1)].data(), bufs[k + 1].data(), BUF_SIZE,
                                        in2_ref[k] == SavedGraph::Step::NO_REF ? nullptr : bufs[(std::size_t)(in2_ref[k] + 1)].data());
            const std::vector<float> &yb = bufs.back();
            for (uint32_t f = 0; f < BUF_SIZE; ++f)
                for (uint32_t c = 0; c < N; ++c) {
                    const float want = y[(f0 + f) * C + c % C];
                    worst = std::max(worst, ulp(yb[f * N + c], want));
                    max_abs = std::max(max_abs, (double)std::fabs(want));
                    max_err = std::max(max_err, (double)std::fabs(yb[f * N + c] - want));
                }
        }
        std::printf("max ulp %d rel err %.3g engines %zu\n", worst, max_err / max_abs, engines.size());
        const bool loose = argc > 6 && std::strcmp(argv[6], "--fir-tolerance") == 0;
        return (loose ? max_err <= 1e-5 * max_abs : worst <= 1) ? 0 : 1;
    } catch (const std::exception &e) {
        std::printf("error: %s\n", e.what());
        return 3;
    }
}
