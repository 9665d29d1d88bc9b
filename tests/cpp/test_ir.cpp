// The C++ impulse-response loader (include/dspfx_ir.hpp): test_ir <file.wav> [--no-resample] prints the taps as hex doubles.
// Built and run by tests/test_cpp_ir.py (g++ only, no library needed).
#include <cstdio>
#include <cstring>

#include "../../include/dspfx_ir.hpp"

int main(int argc, char **argv) {
    if (argc < 2) return 2;
    try {
        const std::vector<double> h = dspfx::load_impulse_response(argv[1], !(argc > 2 && std::strcmp(argv[2], "--no-resample") == 0));
        for (double v : h) std::printf("%a\n", v);
        return 0;
    } catch (const std::exception &e) {
        std::printf("error: %s\n", e.what());
        return 3;
    }
}
