#!/bin/bash
# GPU box: instruction-mix counters of the trace kernel on a bench workload, for librsx or a variant library.
# usage: tools/pmc_sq.sh <workload> "<variants|base>"
R=${GRAFT_REPO_ROOT:-$(pwd)}
WL=${1:-c3}; VARS=${2:-base}
cd /tmp && export TMPDIR=/tmp
for v in $VARS; do
  lib=$R/source_amd/lib/variants/librsx_$v.so; [ "$v" = base ] && lib=$R/source_amd/lib/librsx.so
  OUT=$R/gpurun_out/pmcsq_${WL}_$v
  rm -rf "$OUT"; mkdir -p "$OUT"
  for grp in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM" "SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU GRBM_GUI_ACTIVE SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_INST_CYCLES_SALU"; do
    tag=$(echo "$grp" | tr ' ' '_' | cut -c1-30)
    RSX_LIB=$lib RSX_PIPELINE=1 timeout 200 rocprofv3 --pmc $grp -d "$OUT/$tag" -o k --output-format csv -- python $R/bench.py --child --workload $WL --steps 3 > "$OUT/$tag.log" 2>&1
  done
  echo "== $WL $v"
  python3 - "$OUT" <<'PY'
import csv, glob, collections, sys, os
agg = collections.defaultdict(list)
for f in glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if os.environ.get('KERNEL', 'k_render_trace') in r['Kernel_Name']:
            agg[r['Counter_Name']].append(float(r['Counter_Value']))
for k, v in sorted(agg.items()):
    print("%-28s n=%2d avg=%.5g" % (k, len(v), sum(v) / len(v)))
PY
done
