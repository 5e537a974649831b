#!/bin/bash
# round 5 (GPU box): mesh-free path forms — path tests, then A/B per scene: tree (three waves for the plain form), two waves
# (variant w2), and the forms with the mesh walk (RSX_PATH_MESHES=1)
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R
OUT=$R/gpurun_out/r5_nomesh; mkdir -p $OUT
timeout 1500 python -m pytest tests/test_gpu_parity.py -x -q -k "staged or lambert or dielectric or cornell or prism or importance or furnace or arena or handed or c5 or volume or deferred or host_callback or user_written or path" > $OUT/tests.txt 2>&1
tail -5 $OUT/tests.txt
for cfg in cornell lambert_plain lambert_vol lambert glass prism; do
  echo "== $cfg tree";        timeout 300 python tools/kbench.py 6 $cfg 2>&1 | tail -1
  echo "== $cfg two waves";   RSX_LIB=$R/source_amd/lib/variants/librsx_w2.so timeout 300 python tools/kbench.py 6 $cfg 2>&1 | tail -1
  echo "== $cfg mesh forms";  RSX_PATH_MESHES=1 timeout 300 python tools/kbench.py 6 $cfg 2>&1 | tail -1
done 2>&1 | tee $OUT/kbench.txt
