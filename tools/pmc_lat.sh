#!/bin/bash
# GPU box: average memory-instruction latencies of the trace kernel (SQ_INST_LEVEL_* / SQ_INSTS_*) on a bench workload.
R=${GRAFT_REPO_ROOT:-$(pwd)}
WL=${1:-c3}
OUT=$R/gpurun_out/pmclat_$WL
rm -rf "$OUT"; mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
for grp in "SQ_INST_LEVEL_VMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVE_CYCLES" "SQ_INST_LEVEL_SMEM SQ_INSTS_SMEM SQ_WAIT_INST_ANY SQ_WAIT_ANY" "SQ_INST_LEVEL_LDS SQ_INSTS_LDS SQ_INSTS_FLAT SQ_ACTIVE_INST_ANY"; do
  tag=$(echo "$grp" | tr ' ' '_' | cut -c1-30)
  RSX_PIPELINE=1 timeout 200 rocprofv3 --pmc $grp -d "$OUT/$tag" -o k --output-format csv -- python $R/bench.py --child --workload $WL --steps 3 > "$OUT/$tag.log" 2>&1
done
python3 - "$OUT" <<'PY'
import csv, glob, collections, sys, os
agg = collections.defaultdict(list)
for f in glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if os.environ.get('KERNEL', 'k_render_trace') in r['Kernel_Name']:
            agg[r['Counter_Name']].append(float(r['Counter_Value']))
a = {k: sum(v) / len(v) for k, v in agg.items()}
for k, v in sorted(a.items()):
    print("%-24s %.5g" % (k, v))
for lvl, n in (("SQ_INST_LEVEL_VMEM", ("SQ_INSTS_VMEM_RD", "SQ_INSTS_VMEM_WR")), ("SQ_INST_LEVEL_SMEM", ("SQ_INSTS_SMEM",)), ("SQ_INST_LEVEL_LDS", ("SQ_INSTS_LDS",))):
    if lvl in a:
        d = sum(a.get(x, 0) for x in n)
        if d: print("avg latency %s: %.0f cycles (level counts per 4 cycles?)" % (lvl, a[lvl] / d))
PY
