#!/bin/bash
# round 5 (GPU box): arena blocks reserved per wave — path tests, then both forms on the path workloads
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R
OUT=$R/gpurun_out/r5_arena; mkdir -p $OUT
timeout 1200 python -m pytest tests/test_gpu_parity.py -x -q -k "staged or lambert or dielectric or cornell or prism or importance or furnace or arena or handed or c5 or volume or deferred or host_callback or user_written" > $OUT/tests.txt 2>&1
tail -5 $OUT/tests.txt
for cfg in cornell lambert prism; do
  for wf in 0 1; do
    echo "== $cfg RSX_WAVEFRONT=$wf"
    RSX_WAVEFRONT=$wf timeout 300 python tools/kbench.py 6 $cfg 2>&1 | tail -1
  done
done 2>&1 | tee $OUT/kbench.txt
