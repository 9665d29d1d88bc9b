// fir_kernels.hip -- see fir_kernels.h.  -ffp-contract=off like the rest of the library.
#include "fir_kernels.h"

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "../../include/dspfx.h"

namespace dspfx {

static thread_local std::string g_fir_err;
const char *fir_last_error() { return g_fir_err.c_str(); }

#define FIRCHK(call)                                                              \
    do {                                                                          \
        hipError_t e__ = (call);                                                  \
        if (e__ != hipSuccess) {                                                  \
            g_fir_err = std::string(#call) + ": " + hipGetErrorString(e__);       \
            return e__ == hipErrorOutOfMemory ? DSPFX_ERR_OOM : DSPFX_ERR_HIP;    \
        }                                                                         \
    } while (0)

// ring[(t0 + f) mod R][c] = port value of in[f][c]  (fir.rs:193 push_back, after the
// collect_and_average hop when enabled)
__global__ void __launch_bounds__(256) fir_append_kernel(const float *in, float *ring, uint32_t N, uint32_t nframes,
                                                         uint32_t row0, uint32_t R, int hop, float hop_div,
                                                         const Layout lay) {
    const size_t total = (size_t)N * nframes;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
        const uint32_t f = (uint32_t)(i / N), c = (uint32_t)(i % N);
        float x = in[lay.at(f, c)];
        if (hop) x = (0.0f + x) / hop_div;
        uint32_t r = row0 + f;
        r = r >= R ? r - R : r;
        ring[(size_t)r * N + c] = x;
    }
}

// Exact path: one lane per (frame, channel); sequential f64 accumulation in deque
// order like Iterator::sum (fir.rs:204-206), cast to f32, + 0.0f (the empty `b`
// slice, 208-216), * divisor (222).  General in n0 (covers the warm-up quirk).
__global__ void __launch_bounds__(256) fir_exact_kernel(const float *ring, const double *taps, float *out, uint32_t N,
                                                        uint32_t nframes, uint32_t T, uint32_t R, uint64_t n0,
                                                        float divisor, const Layout lay) {
    const size_t total = (size_t)N * nframes;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
        const uint32_t f = (uint32_t)(i / N), c = (uint32_t)(i % N);
        const uint64_t n = n0 + f;                          // absolute index of this output
        const uint64_t first = n + 1 >= T ? n + 1 - T : 0;  // oldest sample still in the deque
        const uint32_t L = (uint32_t)(n - first + 1);       // deque length (<= T)
        uint32_t r = (uint32_t)(first % R);
        double acc = 0.0;
        for (uint32_t k = 0; k < L; ++k) {
            acc += (double)ring[(size_t)r * N + c] * taps[k];
            r = r + 1 == R ? 0 : r + 1;
        }
        const float a = (float)acc;
        const float val = a + 0.0f;
        out[lay.at(f, c)] = val * divisor;
    }
}

int fir_configure(FirState &s, const double *taps_reversed, uint32_t n_taps, int mode, uint32_t N,
                  uint32_t max_frames) {
    fir_free(s);
    s.T = n_taps;
    s.N = N;
    s.max_frames = max_frames;
    s.mode = mode;
    s.R = n_taps - 1 + max_frames;
    s.pad = max_frames;
    s.n_seen = 0;
    FIRCHK(hipMalloc((void **)&s.ring, (size_t)s.R * N * sizeof(float)));
    FIRCHK(hipMemset(s.ring, 0, (size_t)s.R * N * sizeof(float)));
    FIRCHK(hipMalloc((void **)&s.taps64, (size_t)n_taps * sizeof(double)));
    FIRCHK(hipMemcpy(s.taps64, taps_reversed, (size_t)n_taps * sizeof(double), hipMemcpyHostToDevice));
    std::vector<float> t32((size_t)n_taps + 2 * s.pad, 0.0f);
    for (uint32_t i = 0; i < n_taps; ++i) t32[s.pad + i] = (float)taps_reversed[i];
    FIRCHK(hipMalloc((void **)&s.taps32, t32.size() * sizeof(float)));
    FIRCHK(hipMemcpy(s.taps32, t32.data(), t32.size() * sizeof(float), hipMemcpyHostToDevice));
    const char *k = getenv("DSPFX_FIR_KERNEL");
    s.kernel = k ? atoi(k) : 0;
    return 0;
}

void fir_free(FirState &s) {
    if (s.ring) (void)hipFree(s.ring);
    if (s.taps64) (void)hipFree(s.taps64);
    if (s.taps32) (void)hipFree(s.taps32);
    s.ring = nullptr;
    s.taps64 = nullptr;
    s.taps32 = nullptr;
}

void fir_reset(FirState &s) {
    if (s.ring) (void)hipMemset(s.ring, 0, (size_t)s.R * s.N * sizeof(float));
    s.n_seen = 0;
}

static unsigned grid_for(size_t total) {
    size_t b = (total + 255) / 256;
    if (b > 256 * 32) b = 256 * 32;
    return (unsigned)(b ? b : 1);
}

int fir_process(FirState &s, const float *in, float *out, uint32_t nframes, int hop, float hop_div,
                const Layout &lay, hipStream_t stream) {
    if (nframes > s.max_frames) {
        g_fir_err = "nframes > max_frames";
        return DSPFX_ERR_INVALID;
    }
    const size_t total = (size_t)s.N * nframes;
    const uint32_t row0 = (uint32_t)(s.n_seen % s.R);
    hipLaunchKernelGGL(fir_append_kernel, dim3(grid_for(total)), dim3(256), 0, stream, in, s.ring, s.N, nframes, row0,
                       s.R, hop, hop_div, lay);
    // fir.rs:187-190
    const float divisor = s.mode == DSPFX_FIR_AVERAGE ? 1.0f / (float)s.T : 1.0f;
    hipLaunchKernelGGL(fir_exact_kernel, dim3(grid_for(total)), dim3(256), 0, stream, s.ring, s.taps64, out, s.N,
                       nframes, s.T, s.R, s.n_seen, divisor, lay);
    FIRCHK(hipGetLastError());
    s.n_seen += nframes;
    return 0;
}

// exported state: u64 n_seen, then the T-1 most recent samples [t][N], oldest first
size_t fir_state_bytes(const FirState &s) { return 8 + (size_t)(s.T - 1) * s.N * sizeof(float); }

int fir_state_export(FirState &s, void *host_dst) {
    memcpy(host_dst, &s.n_seen, 8);
    char *dst = (char *)host_dst + 8;
    const size_t row = (size_t)s.N * sizeof(float);
    for (uint32_t k = 0; k + 1 < s.T; ++k) {
        // sample time n_seen - (T-1) + k ; before the start of time => zeros
        const int64_t t = (int64_t)s.n_seen - (int64_t)(s.T - 1) + k;
        if (t < 0) {
            memset(dst + (size_t)k * row, 0, row);
        } else {
            FIRCHK(hipMemcpy(dst + (size_t)k * row, s.ring + (size_t)((uint64_t)t % s.R) * s.N, row,
                             hipMemcpyDeviceToHost));
        }
    }
    return 0;
}

int fir_state_import(FirState &s, const void *host_src) {
    uint64_t seen;
    memcpy(&seen, host_src, 8);
    const char *src = (const char *)host_src + 8;
    const size_t row = (size_t)s.N * sizeof(float);
    FIRCHK(hipMemset(s.ring, 0, (size_t)s.R * row));
    // re-base time so that the imported history ends at ring row T-2
    const uint64_t hist = seen < s.T - 1 ? seen : s.T - 1;
    for (uint64_t k = 0; k < hist; ++k) {
        const uint64_t srow = (s.T - 1) - hist + k;
        FIRCHK(hipMemcpy(s.ring + (size_t)k * s.N, src + (size_t)srow * row, row, hipMemcpyHostToDevice));
    }
    s.n_seen = hist;
    return 0;
}

const char *fir_kernel_name(const FirState &s) { return s.kernel == 1 ? "fir_mfma_f32" : "fir_exact_f64"; }

}  // namespace dspfx
