#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/pmch
rm -rf "$OUT"; mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
python $R/tools/heavy_rect.py
for grp in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU" "SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_ACTIVE_INST_SCA SQ_INSTS_SMEM SQ_INSTS_BRANCH GRBM_GUI_ACTIVE"; do
    tag=$(echo "$grp" | tr ' ' '_' | cut -c1-30)
    timeout 200 rocprofv3 --pmc $grp -d "$OUT/$tag" -o k --output-format csv -- python $R/tools/heavy_rect.py > "$OUT/$tag.log" 2>&1
done
python3 - <<PY
import csv, glob, collections
agg=collections.defaultdict(list)
for f in glob.glob("$OUT/*/k_counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        if 'k_render_trace' in r['Kernel_Name']:
            agg[r['Counter_Name']].append(float(r['Counter_Value']))
for k,v in sorted(agg.items()):
    print("%-28s n=%2d last=%.5g" % (k, len(v), v[-1]))
PY
