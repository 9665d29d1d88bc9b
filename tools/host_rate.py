import sys, time, numpy as np
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
from __graft_entry__ import load_package
import chains
pkg = load_package()
N, B = 1 << 20, 128
eng = pkg.Engine(N, B)
eng.set_chain(chains.chain5(pkg, 24000))
x = np.random.default_rng(0).uniform(-1, 1, (B, N)).astype(np.float32)
def rate(xin, yout, label):
    for _ in range(2): eng.process_host(xin, out=yout)
    t0 = time.perf_counter(); n = 5
    for _ in range(n): eng.process_host(xin, out=yout)
    dt = (time.perf_counter() - t0) / n
    print("%s: %.1f ms per block, %.3e samples/s, %.1f GB/s over the bus (in+out)" % (label, dt * 1e3, N * B / dt, 2 * xin.nbytes / dt / 1e9))
rate(x, np.empty_like(x), "pageable buffers")
px, py = pkg.PinnedArray((B, N)), pkg.PinnedArray((B, N))
px.array[:] = x
rate(px.array, py.array, "pinned, pipelined")

