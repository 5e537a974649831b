#!/usr/bin/env python3
"""Condenses a gpurun_out/prof_<tag>/ directory (tools/profile_c2.sh) into profiles/<tag>_*.csv/json (committed)."""
import collections
import csv
import glob
import json
import os
import shutil
import sys

tag = sys.argv[1] if len(sys.argv) > 1 else "r01"
src = os.path.join("gpurun_out", "prof_" + tag)
dst = "profiles"
os.makedirs(dst, exist_ok=True)
shutil.copy(os.path.join(src, "trace", "c2_kernel_stats.csv"), os.path.join(dst, tag + "_c2_kernel_stats.csv"))
counters = collections.defaultdict(lambda: collections.defaultdict(list))
for f in sorted(glob.glob(os.path.join(src, "pmc_*", "c2_counter_collection.csv"))):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0].replace("void ", "").split("<")[0]
        if "rocclr" not in k:
            counters[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
avg = {k: {c: sum(v) / len(v) for c, v in d.items()} for k, d in counters.items()}
with open(os.path.join(dst, tag + "_c2_pmc_summary.csv"), "w") as f:
    w = csv.writer(f)
    w.writerow(["kernel", "counter", "launches", "average_per_launch"])
    for k, d in sorted(counters.items()):
        for c, v in sorted(d.items()):
            w.writerow([k, c, len(v), "%.6g" % (sum(v) / len(v))])
out = {}
for k, d in avg.items():
    if "FETCH_SIZE" in d and "WRITE_SIZE" in d:
        # rocprofv3 FETCH_SIZE / WRITE_SIZE are in KiB; on gfx950 FETCH_SIZE reads 1/2 of the bytes (MI355X_MICROARCH.md §HBM)
        out[k] = {"fetch_kib_raw": d["FETCH_SIZE"], "write_kib_raw": d["WRITE_SIZE"],
                  "hbm_bytes_per_launch": int(d["FETCH_SIZE"] * 1024 * 2 + d["WRITE_SIZE"] * 1024),
                  "l2_hit_rate": d.get("TCC_HIT_sum", 0) / max(1.0, d.get("TCC_HIT_sum", 0) + d.get("TCC_MISS_sum", 0)),
                  "wave_wait_frac": d.get("SQ_WAIT_ANY", 0) / max(1.0, d.get("SQ_WAVE_CYCLES", 0)),
                  "valu_lane_utilisation": d.get("SQ_THREAD_CYCLES_VALU", 0) / max(1.0, 64 * d.get("SQ_ACTIVE_INST_VALU", 0)),
                  "waves": d.get("SQ_WAVES")}
json.dump({"command": "python bench.py --steps 10 --warmup 2 --no-cpu-baseline", "kernels": out,
           "hbm_bytes_per_launch": out.get("k_render_trace", {}).get("hbm_bytes_per_launch")},
          open(os.path.join(dst, tag + "_pmc_c2.json"), "w"), indent=1)
print(json.dumps(out, indent=1))
