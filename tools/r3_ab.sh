#!/bin/bash
# GPU box: A/B of the packet walk against the per-lane walk on the bench workloads (no PMC children, no CPU baseline).
# usage: tools/r3_ab.sh <tag> "<workloads>" "<RSX_PACKET_MIN_SPP values>"
R=${GRAFT_REPO_ROOT:-$(pwd)}
TAG=${1:-ab}; WLS=${2:-c3}; VALS=${3:-"0 2"}
OUT=$R/gpurun_out/$TAG
mkdir -p "$OUT"
cd "$R"
for wl in $WLS; do
  for v in $VALS; do
    RSX_PACKET_MIN_SPP=$v timeout 300 python bench.py --workload $wl --steps 20 --warmup 3 --no-pmc --no-cpu-baseline > "$OUT/bench_${wl}_p$v.json" 2> "$OUT/bench_${wl}_p$v.err"
    echo "$wl packet_min_spp=$v rc=$? $(python - "$OUT/bench_${wl}_p$v.json" <<'PY'
import json, sys
try:
    d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    print("ms/step %.3f value %.4g kernel_ms %s" % (d["ms_per_step"], d["value"], d.get("roofline", {}).get("kernel_ms")))
except Exception as e:
    print("unreadable", e)
PY
)"
  done
done
