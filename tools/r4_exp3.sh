#!/bin/bash
# round 4 (GPU box): where the packet kernel's time goes (ablation builds: wrong frames, timing only) and the cheaper unit_pixel
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R; OUT=$R/gpurun_out/r4_exp3; mkdir -p $OUT
AB_VARIANTS="${1:-diet1 base abl_trace abl_mesh abl_tris}" bash tools/ab_bench.sh 2>&1 | tee $OUT/ab.txt
