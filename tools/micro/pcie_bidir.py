import torch, time
n = 512 << 20
h_in = torch.empty(n, dtype=torch.uint8).pin_memory(); h_out = torch.empty(n, dtype=torch.uint8).pin_memory()
d_in = torch.empty(n, dtype=torch.uint8, device="cuda"); d_out = torch.empty(n, dtype=torch.uint8, device="cuda")
s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
def run(mode, parts=1, reps=5):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(reps):
        p = n // parts
        for k in range(parts):
            if mode in ("h2d", "both"):
                with torch.cuda.stream(s1): d_in[k*p:(k+1)*p].copy_(h_in[k*p:(k+1)*p], non_blocking=True)
            if mode in ("d2h", "both"):
                with torch.cuda.stream(s2): h_out[k*p:(k+1)*p].copy_(d_out[k*p:(k+1)*p], non_blocking=True)
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / reps
    tot = n * (2 if mode == "both" else 1)
    print("%-5s parts %3d: %.2f ms per 512 MiB (each way), %.1f GB/s combined" % (mode, parts, dt * 1e3, tot / dt / 1e9), flush=True)
for m in ("h2d", "d2h", "both"):
    for parts in (1, 16):
        run(m, parts)
