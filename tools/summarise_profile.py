#!/usr/bin/env python3
"""Condenses a gpurun_out/prof_<tag>_<workload>/ directory (tools/profile.sh) into profiles/<tag>_*.csv/json (committed).
usage: tools/summarise_profile.py <tag> [c3|c2|c4]"""
import collections
import csv
import glob
import json
import os
import shutil
import sys

tag = sys.argv[1] if len(sys.argv) > 1 else "r01"
wl = sys.argv[2] if len(sys.argv) > 2 else "c3"
src = os.path.join("gpurun_out", "prof_%s_%s" % (tag, wl))
dst = "profiles"
os.makedirs(dst, exist_ok=True)
shutil.copy(os.path.join(src, "trace", "k_kernel_stats.csv"), os.path.join(dst, "%s_%s_kernel_stats.csv" % (tag, wl)))
for line in open(os.path.join(src, "trace.log")):
    if line.startswith('{"metric"'):
        open(os.path.join(dst, "%s_%s_bench_under_rocprof.json" % (tag, wl)), "w").write(line)
counters = collections.defaultdict(lambda: collections.defaultdict(list))
for f in sorted(glob.glob(os.path.join(src, "pmc_*", "k_counter_collection.csv"))):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0].replace("void ", "").replace("<false, 0>", "").replace("<true, 0>", "")
        if "rocclr" not in k:
            counters[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
avg = {k: {c: sum(v) / len(v) for c, v in d.items()} for k, d in counters.items()}
with open(os.path.join(dst, "%s_%s_pmc_summary.csv" % (tag, wl)), "w") as f:
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
json.dump({"command": "RSX_PIPELINE=1 rocprofv3 --pmc <group> -- python tools/kbench.py 4 <%s>, one counter group per run" % wl, "kernels": out,
           "hbm_bytes_per_launch": next((v.get("hbm_bytes_per_launch") for k, v in sorted(out.items()) if k.startswith("k_render_trace")), None)},
          open(os.path.join(dst, "%s_pmc_%s.json" % (tag, wl)), "w"), indent=1)
print(json.dumps(out, indent=1))
