#!/bin/bash
# round 6, A/B 2: both children prefetched inside the hand-written descent (in-tree) against one dependent load per step (nopf), and the compiled descent (noasm)
cd ${GRAFT_REPO_ROOT:-.}
mkdir -p gpurun_out/r6_ab2
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "packet or c3_full or c4_full or philox_frame or random_analytic or frames_instanced or csg_demo_world or fused_welford or passes_per_call or auto_batched or frames_c2" 2>&1 | tail -15 > gpurun_out/r6_ab2/tests.txt
cat gpurun_out/r6_ab2/tests.txt
for r in 1 2; do tools/ab.sh "base nopf noasm" "c3full flat" 10 2>&1 | tee -a gpurun_out/r6_ab2/ab.txt; done
