"""Both directions of the bus at once, a [128][N] block cut into channel parts: pitched (hipMemcpy2DAsync, what dspfx_process_host
issues for the frame-major layout) against linear copies of the same bytes."""
import ctypes as C, time
hip = C.CDLL("libamdhip64.so")
N, B = 1 << 20, 128
nbytes = N * B * 4
def chk(rc, what=""):
    assert rc == 0, (what, rc)
h_in, h_out, d_in, d_out = C.c_void_p(), C.c_void_p(), C.c_void_p(), C.c_void_p()
chk(hip.hipHostMalloc(C.byref(h_in), C.c_size_t(nbytes), 0)); chk(hip.hipHostMalloc(C.byref(h_out), C.c_size_t(nbytes), 0))
chk(hip.hipMalloc(C.byref(d_in), C.c_size_t(nbytes))); chk(hip.hipMalloc(C.byref(d_out), C.c_size_t(nbytes)))
C.memset(h_in, 1, nbytes); C.memset(h_out, 0, nbytes)
s1, s2 = C.c_void_p(), C.c_void_p()
chk(hip.hipStreamCreateWithFlags(C.byref(s1), 1)); chk(hip.hipStreamCreateWithFlags(C.byref(s2), 1))
H2D, D2H = 1, 2
hip.hipMemcpy2DAsync.argtypes = [C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t, C.c_size_t, C.c_size_t, C.c_int, C.c_void_p]
hip.hipMemcpyAsync.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int, C.c_void_p]
def run(kind, part, reps=4):
    n_parts = N // part
    hip.hipDeviceSynchronize(); t0 = time.perf_counter()
    for _ in range(reps):
        for p in range(n_parts):
            if kind == "2d":
                off = p * part * 4
                chk(hip.hipMemcpy2DAsync(d_in.value + off, N * 4, h_in.value + off, N * 4, part * 4, B, H2D, s1))
                chk(hip.hipMemcpy2DAsync(h_out.value + off, N * 4, d_out.value + off, N * 4, part * 4, B, D2H, s2))
            else:
                off = p * part * 4 * B
                chk(hip.hipMemcpyAsync(d_in.value + off, h_in.value + off, part * 4 * B, H2D, s1))
                chk(hip.hipMemcpyAsync(h_out.value + off, d_out.value + off, part * 4 * B, D2H, s2))
    hip.hipDeviceSynchronize(); dt = (time.perf_counter() - t0) / reps
    print("%-6s parts of %7d channels: %.2f ms per block, %.1f GB/s combined" % (kind, part, dt * 1e3, 2 * nbytes / dt / 1e9), flush=True)
for part in (32768, 65536, 131072, 262144):
    run("2d", part); run("linear", part)
# --- the same with the pipeline's dependencies: upload (s1) -> event -> a small device op (s3) -> event -> download (s2)
s3 = C.c_void_p(); chk(hip.hipStreamCreateWithFlags(C.byref(s3), 1))
hip.hipEventCreateWithFlags.argtypes = [C.c_void_p, C.c_uint]
def mkev():
    e = C.c_void_p(); chk(hip.hipEventCreateWithFlags(C.byref(e), 2)); return e
evs = [mkev() for _ in range(64)]
hip.hipMemsetAsync.argtypes = [C.c_void_p, C.c_int, C.c_size_t, C.c_void_p]
hip.hipEventRecord.argtypes = [C.c_void_p, C.c_void_p]; hip.hipStreamWaitEvent.argtypes = [C.c_void_p, C.c_void_p, C.c_uint]
def run_dep(kind, part, reps=4, only=None):
    n_parts = N // part
    hip.hipDeviceSynchronize(); t0 = time.perf_counter()
    for _ in range(reps):
        for p in range(n_parts):
            off = p * part * 4
            if only != "d2h":
                chk(hip.hipMemcpy2DAsync(d_in.value + off, N * 4, h_in.value + off, N * 4, part * 4, B, H2D, s1))
            chk(hip.hipEventRecord(evs[2 * p], s1)); chk(hip.hipStreamWaitEvent(s3, evs[2 * p], 0))
            chk(hip.hipMemsetAsync(d_out.value + off, 0, 4096, s3))
            chk(hip.hipEventRecord(evs[2 * p + 1], s3)); chk(hip.hipStreamWaitEvent(s2, evs[2 * p + 1], 0))
            if only != "h2d":
                chk(hip.hipMemcpy2DAsync(h_out.value + off, N * 4, d_out.value + off, N * 4, part * 4, B, D2H, s2))
    hip.hipDeviceSynchronize(); dt = (time.perf_counter() - t0) / reps
    print("with events, %s, parts of %d channels: %.2f ms per block" % (only or "both", part, dt * 1e3), flush=True)
run_dep("2d", 65536); run_dep("2d", 65536, only="d2h"); run_dep("2d", 65536, only="h2d")
def run_only(direction, kind, part, reps=4):
    n_parts = N // part
    hip.hipDeviceSynchronize(); t0 = time.perf_counter()
    for _ in range(reps):
        for p in range(n_parts):
            off = p * part * 4
            if direction == "h2d": chk(hip.hipMemcpy2DAsync(d_in.value + off, N * 4, h_in.value + off, N * 4, part * 4, B, H2D, s1))
            else: chk(hip.hipMemcpy2DAsync(h_out.value + off, N * 4, d_out.value + off, N * 4, part * 4, B, D2H, s2))
    hip.hipDeviceSynchronize(); dt = (time.perf_counter() - t0) / reps
    print("no events, %s only, 2d parts of %d: %.2f ms per block" % (direction, part, dt * 1e3), flush=True)
run_only("h2d", "2d", 65536); run_only("d2h", "2d", 65536)
