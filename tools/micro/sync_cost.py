import time, torch
torch.cuda.init()
x = torch.zeros(64, device="cuda")
def cost(label):
    torch.cuda.synchronize()
    ts = []
    for _ in range(200):
        t0 = time.perf_counter(); torch.cuda.synchronize(); ts.append((time.perf_counter() - t0) * 1e6)
    ts.sort(); print("%-50s idle torch.cuda.synchronize(): p50 %.1f us  p10 %.1f" % (label, ts[100], ts[20]))
cost("default stream only (unused)")
x.sin_(); cost("default stream used")
s1 = torch.cuda.Stream(priority=-1)
with torch.cuda.stream(s1): x.sin_()
cost("+ a high-priority stream used")
s2 = torch.cuda.Stream()
with torch.cuda.stream(s2): x.cos_()
cost("+ a second stream used")
s3 = torch.cuda.Stream()
with torch.cuda.stream(s3): x.cos_()
cost("+ a third stream used")
e = torch.cuda.Event(enable_timing=True)
with torch.cuda.stream(s1):
    x.sin_(); e.record()
t0 = time.perf_counter(); e.synchronize(); t1 = time.perf_counter(); s1.synchronize(); t2 = time.perf_counter()
print("event.synchronize %.1f us, then stream.synchronize %.1f us" % ((t1 - t0) * 1e6, (t2 - t1) * 1e6))
