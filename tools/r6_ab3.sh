#!/bin/bash
# round 6, A/B 3: leaf / pop transitions inside the hand-written block (in-tree) against the first asm form (nopf) and the compiled walk (asm1 = RSX_PKT_ASM=0 of this tree)
cd ${GRAFT_REPO_ROOT:-.}
mkdir -p gpurun_out/r6_ab3
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "packet or c3_full or c4_full or philox_frame or random_analytic or frames_instanced or csg_demo_world or fused_welford or passes_per_call or auto_batched or frames_c2 or flat_1m or mixed_world" 2>&1 | tail -15 > gpurun_out/r6_ab3/tests.txt
cat gpurun_out/r6_ab3/tests.txt
for r in 1 2; do tools/ab.sh "base nopf asm1" "c3full flat c4full" 10 2>&1 | tee -a gpurun_out/r6_ab3/ab.txt; done
