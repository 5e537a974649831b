#!/bin/bash
# round 6, A/B 1: the hand-written descent (in-tree) against the compiled one (variant noasm): packet parity tests, then kernel times
cd ${GRAFT_REPO_ROOT:-.}
mkdir -p gpurun_out/r6_ab1
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "packet or c3_full or c4_full or philox_frame or random_analytic or frames_instanced or csg_demo_world or fused_welford or passes_per_call or auto_batched" 2>&1 | tail -15 > gpurun_out/r6_ab1/tests.txt
cat gpurun_out/r6_ab1/tests.txt
tools/ab.sh "base noasm" "c3full flat c4full" 10 2>&1 | tee gpurun_out/r6_ab1/ab.txt
tools/ab.sh "base noasm" "c3full" 10 2>&1 | tee -a gpurun_out/r6_ab1/ab.txt
