#!/bin/bash
# round 5 (GPU box): the whole GPU suite, the eager-batch stress, the small-pass workloads with and without eager partial batches, single-ray latency
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R
OUT=$R/gpurun_out/r5_check1; mkdir -p $OUT
timeout 1500 python -m pytest tests -x -q -m gpu > $OUT/tests.txt 2>&1
tail -8 $OUT/tests.txt
timeout 600 python tools/r5_eager_repro.py 3 120 2>&1 | tail -6 | tee $OUT/eager_stress.txt
for eager in 1 0; do
  echo "== c2 RSX_EAGER_BATCH=$eager"
  RSX_EAGER_BATCH=$eager timeout 300 python bench.py --workload c2 --no-pmc --no-cpu-baseline 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'])"
done 2>&1 | tee $OUT/c2.txt
timeout 300 python tools/hit_latency.py 2>&1 | tail -2 | tee $OUT/hit_latency.txt
