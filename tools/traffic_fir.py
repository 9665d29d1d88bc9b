#!/usr/bin/env python3
"""profiles/traffic.json entries of config 4 from this round's FIR counter passes (tools/fir_pmc.sh r03 / r03split ->
profiles/r03_fir_pmc.json, profiles/r03_fir_split_pmc.json)."""
import json
import os
import sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tp = os.path.join(ROOT, "profiles", "traffic.json")
t = json.load(open(tp))
ROUND = os.environ.get("DSPFX_ROUND", "r03")
for key, f, kern in (("cfg4:262144:128", ROUND + "_fir_pmc.json", "fir_skew_kernel"), ("cfg4split:262144:128", ROUND + "_fir_split_pmc.json", "fir_split_kernel"),
                     ("cfg4half:262144:128", ROUND + "_fir_halfp_pmc.json", "fir_halfp_kernel")):
    p = os.path.join(ROOT, "profiles", f)
    if not os.path.exists(p):
        print("missing", p, file=sys.stderr)
        continue
    h = json.load(open(p))["hbm"]
    t[key] = {"hbm_bytes_per_launch": h["hbm_bytes_per_launch"],
              "source": "profiles/%s (%s; rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, separate passes, calibrated on the append pass of the same "
                        "run); from the committed PMC pass of this build, not measured in this run" % (f, kern)}
json.dump(t, open(tp, "w"), indent=1)
print(json.dumps(t, indent=1)[:400])
