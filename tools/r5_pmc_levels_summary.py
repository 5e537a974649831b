#!/usr/bin/env python3
"""Per-launch SQ / HBM counters of the last pass in a tools/r5_pmc_levels.sh output directory (runs anywhere: reads the CSVs)."""
import os, sys
OUT = sys.argv[1]
import csv, glob, collections
rows = collections.OrderedDict()
for d in ("sq", "f", "w"):
    for f in glob.glob(os.path.join(OUT, d, "*counter_collection.csv")):
        for r in csv.DictReader(open(f)):
            if 'rocclr' in r['Kernel_Name']: continue
            key = (int(r['Dispatch_Id']) if d == "sq" else None, r['Kernel_Name'][:44])
            rows.setdefault((d, int(r['Dispatch_Id'])), {"name": r['Kernel_Name'][:44]})[r['Counter_Name']] = float(r['Counter_Value'])
sq = [(k[1], v) for k, v in rows.items() if k[0] == "sq"]
ff = {k[1]: v for k, v in rows.items() if k[0] == "f"}
ww = {k[1]: v for k, v in rows.items() if k[0] == "w"}
sq.sort()
# the last pass = everything after the second-last k_accumulate
acc = [i for i, (d, v) in enumerate(sq) if 'k_accumulate' in v["name"]]
start = acc[-2] + 1 if len(acc) > 1 else 0
tot = collections.Counter()
print("%-46s %9s %6s %6s %6s %9s %9s %8s %8s" % ("kernel", "Mcycles", "valu", "lanes", "wait", "vinst(M)", "sinst(M)", "rd GB", "wr GB"))
for d, c in sq[start:]:
    simd_cycles = c["GRBM_GUI_ACTIVE"] / 8.0 * 256 * 4
    f = ff.get(d, {}); w = ww.get(d, {})
    rd = f.get("FETCH_SIZE", 0) * 1024 * 2 / 1e9           # KiB; gfx950 tallies 128-B reads at 64 B (MI355X_MICROARCH.md, HBM section)
    wr = w.get("WRITE_SIZE", 0) * 1024 / 1e9
    print("%-46s %9.3f %6.3f %6.3f %6.3f %9.2f %9.2f %8.3f %8.3f" % (c["name"], c["GRBM_GUI_ACTIVE"] / 8 / 1e6, c["SQ_ACTIVE_INST_VALU"] * 4 / max(simd_cycles, 1),
          c["SQ_THREAD_CYCLES_VALU"] / max(1.0, 64 * c["SQ_ACTIVE_INST_VALU"]), c["SQ_WAIT_ANY"] / max(1.0, c["SQ_WAVE_CYCLES"]), c["SQ_INSTS_VALU"] / 1e6, c["SQ_INSTS_SALU"] / 1e6, rd, wr))
    for k in ("SQ_INSTS_VALU", "SQ_INSTS_SALU", "SQ_ACTIVE_INST_VALU", "SQ_THREAD_CYCLES_VALU", "GRBM_GUI_ACTIVE"): tot[(c["name"], k)] += c[k]
print("per kernel name, summed over the pass:")
for name in sorted({n for n, _ in tot}):
    g = lambda k: tot[(name, k)]
    print("  %-46s Mcycles %9.3f valu %.3f lanes %.3f vinst %.1f M sinst %.1f M" % (name, g("GRBM_GUI_ACTIVE") / 8 / 1e6, g("SQ_ACTIVE_INST_VALU") * 4 / max(1.0, g("GRBM_GUI_ACTIVE") / 8.0 * 1024),
          g("SQ_THREAD_CYCLES_VALU") / max(1.0, 64 * g("SQ_ACTIVE_INST_VALU")), g("SQ_INSTS_VALU") / 1e6, g("SQ_INSTS_SALU") / 1e6))
