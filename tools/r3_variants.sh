#!/bin/bash
# GPU box: bench.py (no PMC, no CPU baseline) on one workload for librsx and a list of variant libraries.
# usage: tools/r3_variants.sh <tag> <workload> "<variant names | base>" [steps]
R=${GRAFT_REPO_ROOT:-$(pwd)}
TAG=${1:-var}; WL=${2:-c3}; VARS=${3:-base}; STEPS=${4:-20}
OUT=$R/gpurun_out/$TAG
mkdir -p "$OUT"
cd "$R"
for v in $VARS; do
  lib=$R/source_amd/lib/variants/librsx_$v.so; [ "$v" = base ] && lib=$R/source_amd/lib/librsx.so
  RSX_LIB=$lib timeout 300 python bench.py --workload $WL --steps $STEPS --warmup 3 --no-pmc --no-cpu-baseline > "$OUT/bench_${WL}_$v.json" 2> "$OUT/bench_${WL}_$v.err"
  echo "$WL $v rc=$? $(python - "$OUT/bench_${WL}_$v.json" <<'PY'
import json, sys
try:
    d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    print("ms/step %.3f value %.4g kernel_ms %s" % (d["ms_per_step"], d["value"], d.get("roofline", {}).get("kernel_ms")))
except Exception as e:
    print("unreadable", e)
PY
)"
done
