#!/bin/bash
# round 6 (GPU box): configs[1]'s pass (1024^2, 1 spp) rendered K passes per library call: cost per pass against K
cd ${GRAFT_REPO_ROOT:-.}
mkdir -p gpurun_out/r6_c2
for K in 1 2 4 8 16 32 64; do
  echo -n "K=$K: "; timeout 600 python bench.py --workload c2k --passes-per-call $K --steps 20 --warmup 3 --no-pmc --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); k=$K; print('%.4g rays/s, %.4f ms per step = %.4f ms per pass; kernel %s x %.3f ms' % (d['value'], d['ms_per_step'], d['ms_per_step']/k, d['roofline'].get('kernel','?')[:48], d['roofline'].get('kernel_ms_per_launch',0)))"
done 2>&1 | tee gpurun_out/r6_c2/per_k.txt
