cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d /root/repo/gpurun_out/hosttrace -o h -- python3 /root/repo/tools/host_rate.py > /root/repo/gpurun_out/hosttrace.log 2>&1
python3 - <<'PY'
import csv, glob
k = glob.glob("/root/repo/gpurun_out/hosttrace/**/*kernel_trace.csv", recursive=True)
m = glob.glob("/root/repo/gpurun_out/hosttrace/**/*memory_copy_trace.csv", recursive=True)
print(k, m)
ev = []
for r in csv.DictReader(open(m[0])):
    ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Direction"], int(r.get("Bytes", r.get("Size", 0)) or 0)))
for r in csv.DictReader(open(k[0])):
    if "chain" in r["Kernel_Name"]:
        ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), "K", 0))
ev.sort()
# the last process_host call: take the last 16 H2D copies of 32 MiB
h2d = [e for e in ev if "HOST_TO_DEVICE" in e[2] and e[3] >= 30 << 20]
last = h2d[-16:]
t0 = last[0][0]
tend = max(e[1] for e in ev if e[0] >= t0)
print("last block: %.2f ms from first upload start to last event end" % ((tend - t0) / 1e6))
for e in ev:
    if e[0] >= t0:
        print("%8.3f -> %8.3f ms  %-22s %5.1f MiB  %.1f GB/s" % ((e[0] - t0) / 1e6, (e[1] - t0) / 1e6, e[2][:22], e[3] / 2**20, e[3] / max(e[1] - e[0], 1)))
PY
