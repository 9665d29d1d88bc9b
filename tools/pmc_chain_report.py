#!/usr/bin/env python3
"""HBM bytes per launch from the counter passes (tools/pmc_chain.sh): FETCH_SIZE and WRITE_SIZE of the config's kernel,
each scaled by the factor that makes the empty-chain kernel of the same pass report the N*B*4 bytes it is known to move in that
direction.  Writes profiles/<round>_pmc_<cfg>.json (round = $DSPFX_ROUND) and profiles/traffic.json."""
import collections
import csv
import glob
import json
import os
ROUND = os.environ.get("DSPFX_ROUND", "r03")
import statistics
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
out_dir = sys.argv[1]
dest = sys.argv[2] if len(sys.argv) > 2 else os.path.join(ROOT, "profiles")     # on a gpurun box: a directory under gpurun_out/
os.makedirs(dest, exist_ok=True)
traffic = {}
for cfg in ("cfg5", "cfg3", "cfg2"):
    rec = {"round": ROUND, "command": "python3 tools/pmc_chain_workload.py %s 40 (under rocprofv3 --kernel-trace --pmc <counter>)" % cfg,
           "units": "counter values are KiB per dispatch (rocprofv3 FETCH_SIZE / WRITE_SIZE)", "counters": {}}
    info = None
    ok = True
    for c in ("FETCH_SIZE", "WRITE_SIZE"):
        log = os.path.join(out_dir, "%s_%s.log" % (cfg, c))
        files = glob.glob(os.path.join(out_dir, "%s_%s" % (cfg, c), "**", "*counter_collection.csv"), recursive=True)
        if not files or not os.path.exists(log):
            ok = False
            continue
        for line in open(log):
            if line.startswith("PMCINFO"):
                _, _, n, b, kern, cal, bps = line.split()
                info = (int(n), int(b), kern, cal, float(bps))
        if info is None:
            ok = False
            continue
        n, b, kern, cal, bps = info
        vals = collections.defaultdict(list)
        durs = collections.defaultdict(list)
        for r in csv.DictReader(open(files[0])):
            if r["Counter_Name"] != c:
                continue
            vals[r["Kernel_Name"]].append(float(r["Counter_Value"]))
            durs[r["Kernel_Name"]].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
        # the two kernels of the workload: the empty chain (all SigList entries -2) and the config's chain kernel
        cal_names = [k for k in vals if "SigList<-2, -2, -2, -2, -2, -2, -2, -2>" in k]
        main_names = [k for k in vals if ("chain_kernel" in k or "chain_ts_kernel" in k) and k not in cal_names]
        if not cal_names or not main_names:
            ok = False
            continue
        main = max(main_names, key=lambda k: len(vals[k]))
        caln = max(cal_names, key=lambda k: len(vals[k]))
        known_kib = n * b * 4 / 1024.0
        skip = 4                                               # first launches: cold TLB / page tables
        raw, cal_raw = statistics.mean(vals[main][skip:]), statistics.mean(vals[caln][skip:])
        factor = known_kib / cal_raw
        rec["counters"][c] = {"kernel": main, "launches": len(vals[main]), "raw_kib_per_launch": raw,
                              "kernel_us_avg_under_the_counter_pass": statistics.mean(durs[main][skip:]),
                              "calibration_kernel": caln, "calibration_raw_kib": cal_raw, "calibration_known_kib": known_kib,
                              "correction_factor": factor, "corrected_bytes_per_launch": raw * factor * 1024.0}
    if not ok or len(rec["counters"]) != 2:
        print(cfg, "incomplete:", rec)
        continue
    n, b, kern, cal, bps = info
    hbm = sum(v["corrected_bytes_per_launch"] for v in rec["counters"].values())
    alg = bps * n * b
    rec.update({"kernel": kern, "channels": n, "frames": b, "bus": "same block, inside the launch (dspfx_process_bus)",
                "hbm_bytes_per_launch": hbm, "algorithmic_bytes_per_launch": alg, "traffic_over_algorithmic": hbm / alg})
    path = os.path.join(dest, "%s_pmc_%s.json" % (ROUND, cfg))
    json.dump(rec, open(path, "w"), indent=1)
    print("%s %s: %.4f GB per launch = %.4f x algorithmic (FETCH x%.4f, WRITE x%.4f)" % (
        cfg, kern, hbm / 1e9, hbm / alg, rec["counters"]["FETCH_SIZE"]["correction_factor"], rec["counters"]["WRITE_SIZE"]["correction_factor"]))
    traffic["%s:%d:%d" % (cfg, n, b)] = {
        "hbm_bytes_per_launch": hbm,
        "source": "profiles/" + ROUND + "_pmc_%s.json (%s; rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, separate passes, each calibrated on the empty-chain "
                  "kernel %s of the same pass); from the committed PMC pass of this build, not measured in this run" % (cfg, kern, cal)}
tp = os.path.join(ROOT, "profiles", "traffic.json")
old = json.load(open(tp)) if os.path.exists(tp) else {}
for k, v in old.items():                                        # FIR entries are produced by tools/fir_pmc.sh
    if k.startswith("cfg4") and k not in traffic:
        traffic[k] = v
json.dump(traffic, open(os.path.join(dest, "traffic.json"), "w"), indent=1)
