#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r6_c2/ktrace; rm -rf $O; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
for K in 4 8 16; do
  timeout 300 rocprofv3 --kernel-trace --stats -d $O/k$K -o t --output-format csv -- python $R/bench.py --workload c2k --passes-per-call $K --steps 20 --warmup 3 --no-pmc --no-cpu-baseline > $O/k$K.log 2>&1
  echo "== K=$K"; f=$(ls $O/k$K/*kernel_stats.csv | head -1); python3 - "$f" <<'PY'
import csv, sys
for r in list(csv.DictReader(open(sys.argv[1])))[:8]:
    print("  %-70s calls %5s avg %9.3f us  %6s%%" % (r["Name"][:70], r["Calls"], float(r["AverageNs"]) / 1e3, r["Percentage"]))
PY
done
