#!/bin/bash
# round 4 (GPU box): ablation timings of the packet kernel through tools/kbench.py (frames are wrong by construction: timing only)
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R; OUT=$R/gpurun_out/r4_exp4; mkdir -p $OUT
for v in base abl_trace abl_mesh abl_wide abl_wide_mesh; do lib=$R/source_amd/lib/variants/librsx_$v.so; [ $v = base ] && lib=$R/source_amd/lib/librsx.so
  for fuse in 1 0; do echo -n "$v fuse=$fuse: "; RSX_LIB=$lib RSX_FUSE=$fuse timeout 200 python tools/kbench.py 6 c3full 2>&1 | tail -1 | cut -c1-300; done; done 2>&1 | tee $OUT/abl.txt
