// The C++ saved-graph importer (include/dspfx_graph.hpp).  Built and run by tests/test_cpp_graph.py (g++ only).
//   test_graph <doc.json> --plan                      print the plan handed to dspfx_graph_set (no device needed)
//   test_graph <doc.json> <x.f32> <y.f32> <channels> <frames>
//        run the graph on the GPU (64 copies of the given channels, process_host, 128-frame blocks) and compare with y:
//        prints "max ulp N"; exit code 0 when N <= 1
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
    if (argc < 3) return 2;
    try {
        using namespace dspfx;
        const SavedGraph g(slurp(argv[1]));
        std::vector<Node> specs;
        std::vector<dspfx_graph_link> links;
        const bool one = g.fused_plan(specs, links);
        if (std::strcmp(argv[2], "--plan") == 0) {
            if (!one) { std::printf("needs cutting\n"); return 0; }
            for (const Node &n : specs)
                std::printf("node %d %d %a %a %a %a %a %a %u\n", n.d.kind, n.d.mode, n.d.params[0], n.d.params[1], n.d.params[2],
                            n.d.params[3], n.d.params[4], n.d.params[5], n.d.delay_len);
            for (const dspfx_graph_link &l : links) std::printf("link %d %d %d\n", l.src, l.dst, l.port);
            return 0;
        }
        if (argc < 6 || !one) return 2;
        const std::vector<float> x = floats(argv[2]), y = floats(argv[3]);
        const uint32_t C = (uint32_t)std::atoi(argv[4]), frames = (uint32_t)std::atoi(argv[5]), N = 64;
        Engine eng(N, BUF_SIZE, 0);
        g.install(eng);
        std::vector<float> xb(BUF_SIZE * N), yb(BUF_SIZE * N);
        int worst = 0;
        for (uint32_t f0 = 0; f0 + BUF_SIZE <= frames; f0 += BUF_SIZE) {
            for (uint32_t f = 0; f < BUF_SIZE; ++f)
                for (uint32_t c = 0; c < N; ++c) xb[f * N + c] = x[(f0 + f) * C + c % C];
            eng.process_host(xb.data(), yb.data(), BUF_SIZE);
            for (uint32_t f = 0; f < BUF_SIZE; ++f)
                for (uint32_t c = 0; c < N; ++c) worst = std::max(worst, ulp(yb[f * N + c], y[(f0 + f) * C + c % C]));
        }
        std::printf("max ulp %d\n", worst);
        return worst <= 1 ? 0 : 1;
    } catch (const std::exception &e) {
        std::printf("error: %s\n", e.what());
        return 3;
    }
}
