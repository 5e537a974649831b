#!/bin/bash
# round 6, A/B 4: the packet walk's cluster short cut (in-tree) against the walk for every unit (nocl)
cd ${GRAFT_REPO_ROOT:-.}
mkdir -p gpurun_out/r6_ab4
timeout 1200 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "packet or c3_full or c4_full or philox_frame or random_analytic or frames_instanced or csg_demo_world or fused_welford or passes_per_call or auto_batched or frames_c2 or flat_1m or mixed_world or frames_csg or edge_semantics or c5_shape or stream_parity or compiled_transitions" 2>&1 | tail -15 > gpurun_out/r6_ab4/tests.txt
cat gpurun_out/r6_ab4/tests.txt
for r in 1 2; do tools/ab.sh "base nocl" "c3full flat c4full" 10 2>&1 | tee -a gpurun_out/r6_ab4/ab.txt; done
