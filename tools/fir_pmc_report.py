#!/usr/bin/env python3
"""Summarise tools/fir_pmc.sh output (rocprofv3 CSVs) into profiles/<round>_fir_pmc.json.

Per counter: the mean over the dispatches of the FIR sweep kernel that ran in the steady state (the last `--last`
dispatches of the run).  FETCH_SIZE / WRITE_SIZE are calibrated on fir_append_kernel, whose traffic is known exactly
(it reads N*B*4 bytes and writes N*B*4 bytes), as MI355X_MICROARCH.md's HBM section prescribes for gfx950.
usage: fir_pmc_report.py <dir> <round> [--kernel fir_skew_kernel] > profiles/<round>_fir_pmc.json"""
import csv, glob, json, os, sys, collections

d, rnd = sys.argv[1], sys.argv[2]
kname = sys.argv[sys.argv.index("--kernel") + 1] if "--kernel" in sys.argv else "fir_skew_kernel"
LAST = 15
N, B, T = 1 << 18, 128, 4096


def counters(sub, kernel):
    out = collections.defaultdict(list)
    dur = []
    path = os.path.join(d, sub, "cfg4_counter_collection.csv")
    if not os.path.exists(path):
        return {}, []
    seen = set()
    for r in csv.DictReader(open(path)):
        if kernel not in r["Kernel_Name"]:
            continue
        out[r["Counter_Name"]].append(float(r["Counter_Value"]))
        if r["Dispatch_Id"] not in seen:
            seen.add(r["Dispatch_Id"])
            dur.append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
    return {k: sum(v[-LAST:]) / len(v[-LAST:]) for k, v in out.items()}, dur[-LAST:]


def trace_avg(kernel):
    path = os.path.join(d, "trace", "cfg4_kernel_trace.csv")
    v = [(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3 for r in csv.DictReader(open(path)) if kernel in r["Kernel_Name"]]
    return sum(v[-LAST:]) / len(v[-LAST:]), len(v)


us, n_disp = trace_avg(kname)
flops = 2.0 * T * N * B
res = {"round": rnd, "config": "cfg4: 262144 channels, 4096-tap FIR, B=128 (bench.py --config cfg4 --steps 20 --warmup 5)",
       "kernel": kname, "dispatches_in_trace": n_disp, "kernel_avg_us_last_%d" % LAST: us,
       "tflops_algorithmic": flops / us / 1e6, "frac_of_157.3": flops / us / 1e6 / 157.3}
m, mdur = counters("pmc_mfma", kname)
if m:
    gui = m["GRBM_GUI_ACTIVE"] / 8.0                       # summed over the 8 XCDs
    dur_us = sum(mdur) / len(mdur)
    res["mfma"] = {"SQ_VALU_MFMA_BUSY_CYCLES": m["SQ_VALU_MFMA_BUSY_CYCLES"], "GRBM_GUI_ACTIVE": m["GRBM_GUI_ACTIVE"],
                   "kernel_us_in_this_pass": dur_us, "shader_clock_GHz": gui / dur_us / 1e3,
                   "mfma_pipe_utilisation": m["SQ_VALU_MFMA_BUSY_CYCLES"] / (gui * 256 * 4),
                   "note": "utilisation = MFMA busy cycles / (active cycles per XCD x 256 CUs x 4 SIMDs)"}
o, _ = counters("pmc_mops", kname)
if o:
    res["instructions"] = o
    if "SQ_INSTS_MFMA" in o:
        res["instructions"]["busy_cycles_per_mfma"] = m["SQ_VALU_MFMA_BUSY_CYCLES"] / o["SQ_INSTS_MFMA"] if m else None
v, _ = counters("pmc_valu", kname)
if v:                                                   # per-wave instruction counts by unit (VERDICT r04 #3a: VALU next to MFMA)
    res.setdefault("instructions", {}).update(v)
    if o and o.get("SQ_INSTS_MFMA") and v.get("SQ_INSTS_VALU"):
        # SQ_INSTS_VALU counts the MFMAs too: what is left is the conversion / masking / address work per matrix instruction
        res["instructions"]["valu_non_mfma_per_mfma"] = (v["SQ_INSTS_VALU"] - o["SQ_INSTS_MFMA"]) / o["SQ_INSTS_MFMA"]
# the append pass of the same run calibrates the two counters: fir_append_kernel reads N*B*4 bytes and writes N*B*4; fir_append2_kernel
# (the packed history of the two-part f16 sweep, round 5) reads N*B*4 and writes 2*N*B*4 + one peak per channel
app = "fir_append2_kernel" if counters("pmc_fetch", "fir_append2_kernel")[0] else "fir_append_kernel"
fa, _ = counters("pmc_fetch", app)
wa, _ = counters("pmc_write", app)
fk, _ = counters("pmc_fetch", kname)
wk, _ = counters("pmc_write", kname)
if fa and wa and fk and wk:
    known = N * B * 4.0
    known_w = known if app == "fir_append_kernel" else 2.0 * known + N * 4.0
    ff, wf = known / fa["FETCH_SIZE"], known_w / wa["WRITE_SIZE"]
    fetch, write = fk["FETCH_SIZE"] * ff, wk["WRITE_SIZE"] * wf
    alg = (4.0 * (T - 1) / B + 4.0) * N * B
    res["hbm"] = {"calibration_kernel": "%s: reads N*B*4 = %d B and writes %d B" % (app, known, known_w),
                  "FETCH_SIZE_bytes_per_count": ff, "WRITE_SIZE_bytes_per_count": wf,
                  "fetch_bytes_per_launch": fetch, "write_bytes_per_launch": write, "hbm_bytes_per_launch": fetch + write,
                  "algorithmic_bytes_of_this_kernel": alg, "traffic_over_algorithmic": (fetch + write) / alg,
                  "whole_block_bytes": fetch + write + known + known_w, "survey_8d_bytes": 140.0 * N * B}
print(json.dumps(res, indent=1))
