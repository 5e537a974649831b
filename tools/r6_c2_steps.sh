#!/bin/bash
# round 6 (GPU box): configs[1] (1024^2, 1 spp per observe()) at 20 and 64 steps under the batching knobs
cd ${GRAFT_REPO_ROOT:-.}
mkdir -p gpurun_out/r6_c2
run() { echo -n "$1 steps=$2: "; env $1 timeout 600 python bench.py --workload c2 --steps $2 --warmup 3 --no-pmc --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('%.4g rays/s, %.4f ms per step; kernel %s x %.3f ms' % (d['value'], d['ms_per_step'], d['roofline'].get('kernel','?')[:44], d['roofline'].get('kernel_ms_per_launch',0)))"; }
for r in 1 2; do
for steps in 20 64; do
  run "RSX_NONE=1" $steps
  run "RSX_EAGER_MIN=8" $steps
  run "RSX_EAGER_MIN=4" $steps
  run "RSX_EAGER_BATCH=0" $steps
  run "RSX_AUTO_BATCH=0" $steps
done; done 2>&1 | tee gpurun_out/r6_c2/knobs.txt
