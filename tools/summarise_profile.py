#!/usr/bin/env python3
"""Copies the judged pieces of a gpurun_out/prof_<tag>_<workload>/ directory (tools/profile.sh) into profiles/ (committed):
<tag>_<wl>_kernel_stats.csv (rocprofv3 --kernel-trace --stats), <tag>_<wl>_bench_under_rocprof.json, <tag>_bench_<wl>.json (the
un-profiled bench line with roofline ceilings and cpu_baseline) and <tag>_pmc_<wl>.json (per-kernel counters of bench.py's PMC runs).
usage: tools/summarise_profile.py <tag> [c3|c2|c4|flat]"""
import glob
import json
import os
import shutil
import sys

tag = sys.argv[1] if len(sys.argv) > 1 else "r02"
wl = sys.argv[2] if len(sys.argv) > 2 else "c3"
src = os.path.join("gpurun_out", "prof_%s_%s" % (tag, wl))
dst = "profiles"
os.makedirs(dst, exist_ok=True)
stats = glob.glob(os.path.join(src, "trace", "**", "*kernel_stats.csv"), recursive=True)
if stats:
    shutil.copy(stats[0], os.path.join(dst, "%s_%s_kernel_stats.csv" % (tag, wl)))
for line in open(os.path.join(src, "trace.log")):
    if line.startswith('{"metric"'):
        open(os.path.join(dst, "%s_%s_bench_under_rocprof.json" % (tag, wl)), "w").write(line)
bench = os.path.join(src, "bench.json")
if os.path.exists(bench) and os.path.getsize(bench):
    line = [l for l in open(bench) if l.startswith('{"metric"')][-1]
    open(os.path.join(dst, "%s_bench_%s.json" % (tag, wl)), "w").write(line)
    d = json.loads(line)
    r = d["roofline"]
    print("ms/step %.3f, value %.4g %s; bound %s (%s); kernel %.3f ms, Welford: %s%s" % (
        d["ms_per_step"], d["value"], d["unit"], r.get("bound"), r.get("frac"), r["kernel_ms"], r.get("welford"),
        "" if r.get("accumulate_kernel_ms") is None else " %.3f ms" % r["accumulate_kernel_ms"]))
pmc = os.path.join(src, "pmc_%s.json" % wl)
if os.path.exists(pmc):
    shutil.copy(pmc, os.path.join(dst, "%s_pmc_%s.json" % (tag, wl)))
