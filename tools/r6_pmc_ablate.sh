#!/bin/bash
# round 6 (GPU box): instruction counts of the packet kernel per launch for a list of (ablation) libraries: tools/r6_pmc_ablate.sh cfg lib...   ("tree" = in-tree)
R=${GRAFT_REPO_ROOT:-$(pwd)}
CFG=$1; shift
OUT=$R/gpurun_out/pmc_ablate_$CFG
rm -rf "$OUT"; mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
for v in "$@"; do
  if [ $v = tree ]; then unset RSX_LIB; else export RSX_LIB=$R/source_amd/lib/variants/librsx_$v.so; fi
  KB_WARM=1 timeout 300 rocprofv3 --pmc SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_SALU SQ_INSTS_SMEM GRBM_GUI_ACTIVE -d "$OUT/${v}_a" -o k --output-format csv -- python $R/tools/kbench.py 2 $CFG > "$OUT/${v}_a.log" 2>&1
  KB_WARM=1 timeout 300 rocprofv3 --pmc SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_INSTS_BRANCH SQ_ACTIVE_INST_SCA SQ_WAVE_CYCLES SQ_THREAD_CYCLES_VALU SQ_BUSY_CYCLES -d "$OUT/${v}_b" -o k --output-format csv -- python $R/tools/kbench.py 2 $CFG > "$OUT/${v}_b.log" 2>&1
  KB_WARM=1 timeout 300 python $R/tools/kbench.py 5 $CFG 2>/dev/null | tail -1 | cut -c1-140 > "$OUT/${v}_time.txt"
done
python3 - "$@" <<PY | tee "$OUT/summary.txt"
import csv, glob, collections, sys, json
units = None
print("%-14s %9s %8s %8s %7s %7s %7s %7s %7s  %6s %6s" % ("lib", "ms", "VALU/u", "SALU/u", "SMEM/u", "BR/u", "LDS/u", "VMEM/u", "lanes", "Vbusy", "Sbusy"))
for v in sys.argv[1:]:
    c = collections.Counter(); n = collections.Counter()
    for f in glob.glob("$OUT/%s_[ab]/*counter_collection.csv" % v):
        for r in csv.DictReader(open(f)):
            if 'k_render_trace' not in r['Kernel_Name']: continue
            c[r['Counter_Name']] += float(r['Counter_Value']); n[(r['Counter_Name'], r['Dispatch_Id'])] = 1
    launches = collections.Counter(k for k, _ in n)
    for k in c: c[k] /= max(1, launches[k])
    try: ms = json.loads(open("$OUT/%s_time.txt" % v).read().strip() + ('' if open("$OUT/%s_time.txt" % v).read().strip().endswith('}') else '"}'))["trace_ms"]
    except Exception:
        import re; m = re.search(r'"trace_ms": ([0-9.]+)', open("$OUT/%s_time.txt" % v).read()); ms = float(m.group(1)) if m else float('nan')
    u = 4194304.0
    g = c["GRBM_GUI_ACTIVE"] or 1
    print("%-14s %9.3f %8.0f %8.0f %7.0f %7.0f %7.0f %7.0f %7.3f  %6.3f %6.3f" % (v, ms, c["SQ_INSTS_VALU"] / u, c["SQ_INSTS_SALU"] / u, c["SQ_INSTS_SMEM"] / u, c["SQ_INSTS_BRANCH"] / u,
          c["SQ_INSTS_LDS"] / u, (c["SQ_INSTS_VMEM_RD"] + c["SQ_INSTS_VMEM_WR"]) / u, c["SQ_THREAD_CYCLES_VALU"] / max(1, c["SQ_ACTIVE_INST_VALU"]) / 64 / 4 * 4,
          c["SQ_ACTIVE_INST_VALU"] * 4 / (g * 128), c["SQ_ACTIVE_INST_SCA"] * 4 / (g * 128)))
PY
